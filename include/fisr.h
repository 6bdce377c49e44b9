/* libfisr_hip.so -- C-ABI of the MI355X-native FISRnet inference hot path.
 *
 * The reference (JihyongOh/FISR, Python + TensorFlow 1.13) has no FFI layer; the hot
 * path sits behind three Python seams.  Each entry point below
 * names the reference interface it replaces (paths relative to the reference root):
 *
 *   graph seam   FISRnet.model(img, sf) -> (pred_l1, pred_l2, pred_l3)   FISRnet.py:73-173
 *                executed by sess.run(test_Pred, feed_dict=...)          FISRnet.py:871-872, 1048-1049
 *   weight seam  FISRnet.load(checkpoint_dir) / tf.train.Saver.restore   FISRnet.py:751-753, 1101-1115
 *   warp         warp_flow() + YUV2RGB/RGB2YUV                           FISR_tfoptflow/FISR_for_video_warp_img_with_flo.py:35-67,112-129
 *   input pack   normalise / clip / concat                               FISRnet.py:828-843
 *   output       clip, uint8 truncation, YUV2RGB_matlab                  FISRnet.py:883,903-909; utils.py:106-115
 *
 * Conventions: plain C, no torch types.  Every function returns 0 on success or a
 * negative FISR_E* code; the message is available from fisr_last_error() (never
 * throws across the ABI).  All tensor pointers are DEVICE pointers owned by the
 * caller (e.g. torch.Tensor.data_ptr()) unless the name says `host`.  `stream` is a
 * hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); calls are
 * asynchronous on it.  One ctx per device; calls on one ctx are not re-entrant.
 * Layout is NHWC everywhere, exactly as the reference feeds TensorFlow.
 */
#ifndef FISR_H
#define FISR_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: the FISR_API entries below are its whole dynamic surface
 * (tests/test_host.py asserts that `nm -D` shows exactly these). */
#ifndef FISR_API
#define FISR_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fisr_ctx fisr_ctx;

enum {
  FISR_OK = 0,
  FISR_EINVAL = -1,   /* bad argument (shape not multiple of 32, null pointer, ...) */
  FISR_ESTATE = -2,   /* wrong call order (forward before finalize, ...) */
  FISR_EMISSING = -3, /* a required variable was never set */
  FISR_EHIP = -4,     /* HIP runtime error */
  FISR_ENOMEM = -5    /* workspace too small */
};

/* Arithmetic the conv kernels compute in (activations are stored in the same type).
 * ENGINES a caller selects with fisr_finalize_weights: FISR_PREC_F32W4 (fp32, the default engine), FISR_PREC_F32 (fp32, every conv on
 * the direct kernel: the exact reference), FISR_PREC_BF16X3, FISR_PREC_F16F8, FISR_PREC_F16, FISR_PREC_MIXED.
 * DIAGNOSTICS ids -- FISR_PREC_F32W as an engine, FISR_PREC_F16R, FISR_PREC_F16F8R, FISR_PREC_MIXEDR: A/B engines that run a superseded
 * kernel everywhere.  fisr_finalize_weights refuses them (FISR_EINVAL) unless the library is a -DFISR_DIAG build (fisr_version() then
 * ends in "DIAG").  At OP level (fisr_op_conv3x3 ...) the same ids stay valid: there they name a kernel the product still ships for
 * the layers its successors do not take (F(2x2) on small maps; the register-staged direct kernel for heads and narrow layers). */
enum {
  FISR_PREC_F32 = 0, /* fp32 activations/weights, v_mfma_f32_32x32x2_f32 (exact fp32, fmaf chain) */
  FISR_PREC_F16 = 1, /* fp16 activations/weights, v_mfma_f32_32x32x16_f16, fp32 accumulate */
  FISR_PREC_BF16X3 = 2, /* split bf16: every value kept as hi+lo (2 x bf16, ~2^-18 relative), each
                          product = 3 x v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi), fp32
                          accumulate: fp32-grade results at 16/3 x the fp32 MFMA rate.  Activation
                          tensors use the split layout: per pixel, per 16 channels, 16 bf16 hi
                          (32 B) then 16 bf16 lo (32 B). */
  FISR_PREC_F32W = 4, /* fp32 activations/weights like FISR_PREC_F32; the convolutions with Cout % 64 == 0 (132 of the
                         138) run Winograd F(2x2,3x3) minimal filtering in fp32 -- transforms in fp32 adds, products on
                         v_mfma_f32_32x32x2_f32 with fp32 accumulation, 16 instead of 36 multiplies per 2x2 outputs --
                         the algorithm cuDNN uses for fp32 3x3 convolutions under the reference's TF 1.13; the rest
                         (3/6-channel heads) use the direct kernel.  Results differ from FISR_PREC_F32 by fp32
                         rounding only (different summation order).  As an ENGINE (F(2x2) on every map: round 2's headline) it is
                         a DIAGNOSTICS id since r06 (-DFISR_DIAG builds); at op level it names the F(2x2) kernel, which the
                         FISR_PREC_F32W4 engine runs on the small maps, and fisr_pwc_finalize_precision accepts it. */
  FISR_PREC_MIXED = 5, /* engine only (fisr_finalize_weights; not an op-level precision): FISR_PREC_F16 everywhere
                          except at the full and the half resolution of level 3 -- its first two encoder levels, its last
                          two decoder levels and both heads --, which keep a split format (FISR_PREC_F16F8).  The
                          activation format changes twice, on 1/16-size tensors.  PSNR shift of the SR channel against
                          the fp64 oracle <= 0.005 dB on the three weight sets of tests/, 0.007 dB on the worst of 30
                          full-size windows (all-fp16: 0.026 dB, outside north_star's +-0.02 dB). */
  FISR_PREC_F16F8 = 3, /* fp16 + fp8 split: x ~ h + l8*2^-14 (h = fp16(x)); per pixel and 16 channels
                          16 x fp16 h (32 B), then per 8 channels 8 x fp8-e4m3 l8 and 8 x fp8-e4m3 copy of h (2 x 16 B; r05:
                          the fp8 fields are interleaved per 8 channels).
                          Product = a_h*w_h (v_mfma_f32_32x32x16_f16) + both cross terms in ONE
                          block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 per pair of taps: ~2^-15
                          relative per product at 2.1 instead of 3 MFMA-units (fp32 accumulate).  Values beyond
                          the fp16 range saturate to +-65504 when stored (no inf).  r05: the convolutions with Cout % 64 == 0
                          run on the persistent LDS-DMA kernel (conv3x3_dma_fs.h: same products, another summation order). */
  FISR_PREC_F16R = 6,  /* DIAGNOSTICS (engine: -DFISR_DIAG builds only).  FISR_PREC_F16 arithmetic and tensors on round 1's register-staged direct kernel everywhere (conv3x3.h);
                          FISR_PREC_F16 itself runs the convolutions with Cout > 32 on the LDS-DMA kernel (conv3x3_dma.h: same
                          products, another summation order inside a chunk -> results agree to fp32 rounding).  For A/B runs. */
  FISR_PREC_MIXEDR = 7, /* DIAGNOSTICS (-DFISR_DIAG builds only).  Engine only: FISR_PREC_MIXED with FISR_PREC_F16R for its fp16 stages (A/B runs) */
  FISR_PREC_F32W4 = 8, /* fp32 activations/weights and arithmetic like FISR_PREC_F32W, with Winograd F(4x4,3x3) (conv3x3_wf4.h: 36
                          multiplies per 4x4 outputs on v_mfma_f32_16x16x4_f32, a quarter of the direct algorithm's) for the
                          convolutions with Cout % 64 == 0 on maps of at least 48 x 64 pixels and on 512-channel maps (op level: on
                          every map); smaller maps and the rest as in FISR_PREC_F32W; the 2x2 max pooling behind an encoder level
                          is a second store of that level's last convolution.  The F(4,3) transforms are worse conditioned: ~2e-5
                          instead of ~2e-6 per convolution against float64 -- through the network it does not add up (1.5e-6 on the
                          full tile, the F(2x2) engine's figure).  What fisrnet.py's "fp32" selects since round 3. */
  FISR_PREC_F16F8R = 9 /* DIAGNOSTICS (engine: -DFISR_DIAG builds only).  FISR_PREC_F16F8 arithmetic and tensors on round 1's register-staged direct kernel everywhere (conv3x3.h).
                          For A/B runs, as FISR_PREC_F16R is for fp16. */
};

/* flags of fisr_op_conv3x3 */
enum {
  FISR_CONV_RELU_IN = 1,  /* conv(relu(x))           ops.py:41-42 */
  FISR_CONV_RELU_OUT = 2, /* relu(conv(x) [+ res])   ops.py:52,62,70,75 */
  FISR_CONV_D2S = 4,      /* tf.depth_to_space(y, 2) fused into the store, FISRnet.py:99 */
  FISR_CONV_UP2_IN = 8    /* conv(resize_images(x, 2x, BILINEAR)), ops.py:69-70 (Dec_level_res): in0 is the HALF-resolution map
                             [n, h/2, w/2, c0] and is enlarged on its way into the kernel.  fisr_op_conv3x3 takes it where the
                             F(4x4) Winograd kernel does the work (FISR_PREC_F32W4; h, w even; cout % 64 == 0; c0 % 8 == 0; no
                             in1 / res / RELU_IN / D2S) and returns FISR_EINVAL elsewhere: compose fisr_op_upsample2 +
                             fisr_op_conv3x3 there, as the engines do */
};

FISR_API const char* fisr_version(void);

/* ---- context: owns the (re-packed) weights; replaces the tf.Session + variables ---- */
FISR_API int fisr_create(fisr_ctx** out, int device_id);
FISR_API void fisr_destroy(fisr_ctx* ctx);
/* ctx may be NULL: returns the last error of the calling thread. */
FISR_API const char* fisr_last_error(const fisr_ctx* ctx);

/* Weight seam (replaces saver.restore, FISRnet.py:1108).  `tf_var_name` is the
 * reference's variable name, e.g. "FISRnet/level_3/dec/level_0/resize/w"; `host` is the
 * float32 tensor in TF layout (HWIO for .../w, [Cout] for .../b); copied.  Names outside
 * the 276 FISRnet variables (Adam slots, global step) are ignored with return 1. */
FISR_API int fisr_set_weight(fisr_ctx* ctx, const char* tf_var_name, const float* host,
                    const int64_t* shape, int rank);
/* Re-pack for the MFMA kernels and upload.  Fails with FISR_EMISSING (naming the first
 * absent variable) unless all 276 tensors were set. */
FISR_API int fisr_finalize_weights(fisr_ctx* ctx, int precision);
FISR_API int fisr_num_variables_set(const fisr_ctx* ctx);

/* Graph seam.  in: [n,h,w,29] float32; h % 32 == 0 and w % 32 == 0 (FISRnet.py:820-824).
 * out_l3 [n,2h,2w,9] float32 (required); out_l2 [n,h,w,9], out_l1 [n,h/2,w/2,9] optional
 * (NULL to skip the copy-out).  workspace: caller-owned device scratch of at least
 * fisr_workspace_bytes(ctx,n,h,w). */
FISR_API size_t fisr_workspace_bytes(const fisr_ctx* ctx, int n, int h, int w);
FISR_API int fisr_forward(fisr_ctx* ctx, const float* in_nhwc29, int n, int h, int w,
                 float* out_l3, float* out_l2, float* out_l1,
                 void* workspace, size_t workspace_bytes, void* stream);

/* The graph seam with the input assembly folded in (r04): the same forward, but the level inputs are built straight from the
 * windows' source planes -- FISRnet.py:828-843 (normalise + channel order), :853-857 (the tile's rectangle) and :81,112-113,144
 * (sub-sample + concat) in one kernel per level; the [1,h,w,29] tensor of fisr_pack_input, its per-tile slices and their read-back
 * are never written.  Item i is the h x w rectangle at (y0, x0) of a window given by its eleven planes (what fisr_pack_input
 * takes): frames [h0,w0,3] uint8 YUV, flows [h0,w0,2] float32 px (8-byte aligned), warps [h0,w0,3] float32 0..255, all device
 * pointers with row pitch w0.  Results are bit-identical to fisr_pack_input -> slice -> fisr_forward on the same n tiles;
 * n <= FISR_MAX_SRC_ITEMS per call (the items travel as kernel arguments); workspace as for fisr_forward(n, h, w). */
#define FISR_MAX_SRC_ITEMS 16
typedef struct fisr_src_item {
  const uint8_t* frames[3];
  const float* flows[4];
  const float* warps[4];
  int y0, x0;
} fisr_src_item;
FISR_API int fisr_forward_frames(fisr_ctx* ctx, const fisr_src_item* items, int n, int h0, int w0, int h, int w,
                        float* out_l3, float* out_l2, float* out_l1,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Per-kernel timing (HIP events on `stream` around every launch of the next forwards).
 * Off by default; bench.py uses it for the roofline figure of the dominant kernel. */
FISR_API int fisr_profile_enable(fisr_ctx* ctx, int on);
FISR_API int fisr_profile_reset(fisr_ctx* ctx);
/* Fills up to `cap` entries; returns the number of kernel classes.  name[i] points into
 * ctx-owned storage.  flops = algorithmic 2*9*Cin*Cout*pixels summed over launches. */
FISR_API int fisr_profile_read(fisr_ctx* ctx, int cap, const char** name, double* total_ms,
                      int64_t* launches, double* flops, double* bytes);

/* ---- glue kernels (HBM-bound) ---- */

/* Frame warp, one direction (warp script :112-129): src_yuv [h,w,3] float32 0..255 (the
 * OTHER frame of the pair), flow [h,w,2] float32 pixels, flow_scale 0.5 (:122,126),
 * quantized!=0 reproduces cv2.remap's 1/32-px fixed-point coordinates.
 * dst_yuv [h,w,3] float32 0..255 (what the reference stores in the .mat file). */
FISR_API int fisr_warp(const float* src_yuv, const float* flow, float flow_scale, int h, int w,
              int quantized, float* dst_yuv, void* stream);

/* Input assembly for window s (FISRnet.py:828-843): frames_u8 [3][h0,w0,3] uint8 YUV
 * given as three device pointers, flow4 [4][h0,w0,2] float32 px and warp4 [4][h0,w0,3]
 * float32 0..255 given as four pointers each; crops to h x w (top-left), normalises
 * (/255 clip01; /96/2 clip+-1; /255 clip01) and writes [1,h,w,29] float32. */
FISR_API int fisr_pack_input(const uint8_t* const* frames3, const float* const* flow4,
                    const float* const* warp4, int h0, int w0, int h, int w,
                    float* out_nhwc29, void* stream);

/* Output post-processing (FISRnet.py:883,903-909): pred [h,w,9] float32 -> clip[0,1],
 * yuv_u8 [h,w,9] = uint8(x*255) (truncation), rgb_u8 [3][h,w,3] = YUV2RGB_matlab(yuv)
 * truncated (either may be NULL). */
FISR_API int fisr_unpack_output(const float* pred_hw9, int h, int w, uint8_t* yuv_u8, uint8_t* rgb_u8,
                       void* stream);

/* Copy the trimmed interior of a tile prediction into the full frame
 * (trim_patch_boundary utils.py:138-159 + the stitch at FISRnet.py:879-880).
 * tile [th,tw,9] -> full[fh,fw,9] at (dst_y,dst_x), taking rows/cols from (src_y,src_x),
 * size (ch,cw). */
FISR_API int fisr_stitch(const float* tile, int th, int tw, int src_y, int src_x, int ch, int cw,
                float* full, int fh, int fw, int dst_y, int dst_x, void* stream);

/* Sum of squared error of a prediction against uint8 ground truth, in double, as the
 * reference's PSNR sees it (utils.py:23-26 over FISRnet.py:828-831,883):
 * sum((gt_u8/255.0 - clip(pred,0,1))^2).  The result is written to *out_host after the
 * stream is synchronised. */
FISR_API int fisr_sse_vs_u8(const float* pred, const uint8_t* gt_u8, size_t count, double* out_host,
                   void* stream);

/* SSIM the way the reference's SSIM_PIL.compare_ssim computes it on two uint8 images
 * (FISRnet.py:890-891): non-overlapping 7x7 tiles per channel, C1=(0.01*255)^2, C2=(0.03*255)^2,
 * unbiased variance, mean over tiles and the 3 channels.  a, b: [h,w,cstride] uint8, the frame's
 * three channels start at channel `coff`.  Result in *out_host after synchronising the stream. */
FISR_API int fisr_ssim_u8(const uint8_t* a, const uint8_t* b, int h, int w, int cstride, int coff,
                 double* out_host, void* stream);

/* ---- op-level entry points (parity tests of the individual kernels) ---- */

/* y = conv3x3_SAME(concat(in0,in1)) + b [+ res], ops.py:7-11.  Activations are float32
 * (FISR_PREC_F32), float16 (FISR_PREC_F16), split-bf16 (FISR_PREC_BF16X3) or fp16+fp8 (FISR_PREC_F16F8) device tensors; in1/res may be NULL;
 * c0 (and c1) must be multiples of 16 (f32, bf16x3, f16f8) / 32 (f16) channels; w_host [3,3,c0+c1,cout]
 * and b_host [cout] are HOST float32.  With FISR_CONV_D2S, cout % 4 == 0 and out is
 * [n,2h,2w,cout/4].  out_f32 != 0 stores float32 regardless of precision.
 * One output image (h * w * cout elements of the activation type) must stay below 4 GB: the kernels address it with
 * 32-bit byte offsets (FISR_EHIP otherwise; the 2x2 tiles of a 1080p frame are 0.55 GB at most).
 * Synchronous (packs and uploads the weights on every call). */
FISR_API int fisr_op_conv3x3(const void* in0, int c0, const void* in1, int c1, const float* w_host,
                    const float* b_host, int cout, const void* res, void* out, int n, int h,
                    int w, int flags, int precision, int out_f32, void* stream);
/* The last convolution of an encoder level with tf.nn.max_pool(2x2, stride 2) (ops.py:52-54, Enc_level_res) as a SECOND store of its
 * epilogue: out [n,h,w,cout] as fisr_op_conv3x3 writes it, pool_out [n,h/2,w/2,cout] = its 2x2 maxima.  FISR_PREC_F32W4 (the
 * F(4x4) Winograd kernel: a 4x4 tile holds whole pooling windows; needs a residual input, even h / w, no relu-on-load / d2s /
 * fused bilinear) and, r04, FISR_PREC_BF16X3 / FISR_PREC_F16F8 (the direct kernel's record store: a wave's row pair x a lane pair is
 * a pooling window; even h / w, cout % 16 == 0, no d2s / fused bilinear; residual and relu-on-load optional); anything else is
 * FISR_EINVAL. */
FISR_API int fisr_op_conv3x3_pool(const void* in0, int c0, const void* in1, int c1, const float* w_host, const float* b_host,
                         int cout, const void* res, void* out, void* pool_out, int n, int h, int w, int flags, int precision, void* stream);
FISR_API int fisr_op_maxpool2(const void* in, void* out, int n, int h, int w, int c, int precision, void* stream);
FISR_API int fisr_op_upsample2(const void* in, void* out, int n, int h, int w, int c, int precision, void* stream);
/* The input of a level's first convolution (FISRnet.py:81, 112-113, 144): img [n,h,w,29] sub-sampled by s (1 | 2 | 4 = the legacy
 * BICUBIC resize at an integer factor, x[:, ::s, ::s, :]) ++ pred [n,h/s,w/s,9] (nullable: level 1) ++ zeros up to cpad (% 16) channels,
 * float32 [n,h/s,w/s,cpad]: the glue kernels of the engine's `prep` step, for bit-exact tests. */
FISR_API int fisr_op_prep_level_input(const float* img, const float* pred, float* out, int n, int h, int w, int s, int cpad, void* stream);

/* Micro-benchmark of one conv shape (diagnostics; not on the product path): runs `iters`
 * launches of the conv kernel on self-allocated buffers and returns the mean microseconds per
 * launch in *out_us. */
FISR_API int fisr_bench_conv(int precision, int n, int h, int w, int cin, int cout, int flags, int with_res,
                    int iters, double* out_us);

/* ---- multi-GPU exchange: thin RCCL wrappers (north_star: "2K->4K tiles shard across the 8 GPUs of one
 * node with RCCL all-gather of border halos over xGMI"; the reference itself is single-GPU).
 * One communicator per process / GPU.  librccl is opened lazily (dlopen) on the first call, so a
 * single-GPU user of this library has no RCCL dependency.  Python hosts under torch.distributed use its
 * "nccl" backend instead (fisr_amd/dist.py); these are for hosts without torch. ---- */
typedef struct fisr_comm fisr_comm;
#define FISR_COMM_ID_BYTES 128
/* Rank 0 creates the id (host buffer of FISR_COMM_ID_BYTES) and hands it to the other ranks out of band. */
FISR_API int fisr_comm_unique_id(void* id_out);
FISR_API int fisr_comm_init(fisr_comm** out, const void* id, int nranks, int rank, int device_id);
FISR_API int fisr_comm_rank(const fisr_comm* comm);
FISR_API int fisr_comm_size(const fisr_comm* comm);
/* recv[r * bytes_per_rank ...] = send of rank r (device buffers; asynchronous on stream): the gather of
 * halo strips before a tile-parallel forward and of output tiles after it. */
FISR_API int fisr_comm_allgather(fisr_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
/* Exchange `bytes` with one neighbour (send and receive fused in one group; peer may be the own rank). */
FISR_API int fisr_comm_sendrecv(fisr_comm* comm, const void* send, void* recv, size_t bytes, int peer, void* stream);
FISR_API void fisr_comm_destroy(fisr_comm* comm);

/* ---- on-GPU optical flow (cfg5 of BASELINE.json): PWC-Net-large as
 * `FISR_for_video_Compute_Flow` runs it before FISRnet (main.py:207-211;
 * FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:84-147 over FISR_tfoptflow/model_pwcnet.py:1525-1593 with
 * use_dense_cx, use_res_cx, pyr_lvls 6, flow_pred_lvl 2).  Weight seam: the TF variable names of the reference graph
 * ('pwcnet/featpyr/conv1a/kernel' [3,3,3,16] HWIO, '.../bias', 'pwcnet/predict_flow/conv6_0/kernel', 'pwcnet/ctxt/dc_conv21/kernel',
 * 'pwcnet/upsample/up_feat3/kernel' [4,4,2,Cin], ...): what `pwcnet.ckpt-595000` (script :31) holds. ---- */
typedef struct fisr_pwc fisr_pwc;
FISR_API int fisr_pwc_create(fisr_pwc** out, int device_id);
FISR_API void fisr_pwc_destroy(fisr_pwc* ctx);
FISR_API const char* fisr_pwc_last_error(const fisr_pwc* ctx);
/* the 182 variables the inference graph needs: count, then name/shape of variable i (returns the rank) */
FISR_API int fisr_pwc_num_variables(void);
FISR_API int fisr_pwc_variable(int i, const char** name, int64_t* shape4);
/* host float32 tensor in TF layout; unknown names (optimizer slots, ...) are ignored with return 1 */
FISR_API int fisr_pwc_set_weight(fisr_pwc* ctx, const char* tf_var_name, const float* host, const int64_t* shape, int rank);
FISR_API int fisr_pwc_finalize(fisr_pwc* ctx);   /* FISR_EMISSING names the first absent variable; = ..._precision(ctx, FISR_PREC_F32W) */
/* precision FISR_PREC_F32W: float32 tensors and arithmetic (dense layers on the F(2x2,3x3) Winograd kernel); FISR_PREC_F32W4: the
 * same engine with the F(4x4,3x3) kernel's GENERAL instantiation where the (sub-)image of a layer is at least 48 x 64 (r04).
 * FISR_PREC_F16 (cfg5 of
 * BASELINE.json, "bf16"-class 16-bit arithmetic): fp16 feature tensors, fp32 accumulation, float32 flows (flow heads, refinement
 * sums, what is handed to the next level and to the caller); dense layers on the LDS-DMA kernel. */
FISR_API int fisr_pwc_finalize_precision(fisr_pwc* ctx, int precision);
/* The script's whole loop (:104-141) in one call: nframes YUV uint8 frames [h,w,3] (device pointers) -> flows
 * [nframes-1, 2, h, w, 2] float32 LR pixels (pair fr: [0] = fr -> fr+1, [1] = fr+1 -> fr; the layout of the script's 5-D .flo).
 * Every frame is pre-processed and its feature pyramid extracted once, the 2 (nframes-1) directions go through the decoder as
 * batches of up to 8. */
FISR_API size_t fisr_pwc_flow_stack_workspace_bytes(const fisr_pwc* ctx, int nframes, int h, int w);
FISR_API int fisr_pwc_flow_stack(fisr_pwc* ctx, const uint8_t* const* yuv_frames, int nframes, int h, int w, float* flows, void* workspace,
                        size_t workspace_bytes, void* stream);
/* One iteration of the script's loop (:118-140): two YUV uint8 frames [h,w,3] (device) -> YUV->RGB, x2 scikit-image
 * up-resize, uint8 truncation, /255, pad to 64 (adapt_x, model_pwcnet.py:371-411), the network in both directions,
 * x4 bilinear * 4 (:1587-1590), crop, anti-aliased scikit-image down-resize, / 2 -> flow_ab, flow_ba [h,w,2] float32
 * LR pixels (device): pred[fr, 0] and pred[fr, 1] of the script's 5-D .flo. */
FISR_API size_t fisr_pwc_flow_workspace_bytes(const fisr_pwc* ctx, int h, int w);
FISR_API int fisr_pwc_flow_pair(fisr_pwc* ctx, const uint8_t* yuv_a, const uint8_t* yuv_b, int h, int w, float* flow_ab,
                       float* flow_ba, void* workspace, size_t workspace_bytes, void* stream);
/* The network alone (model_pwcnet.py:1525-1593) on a prepared pair: im [2,H,W,4] device float32 (image a, image b;
 * RGB/255 and a zero 4th channel; H, W multiples of 64).  flow_pred [2,H,W,2] (a->b, b->a; nullable); pyr: 10 nullable
 * device pointers, refined flows of levels 6..2 [H/2^l, W/2^l, 2] for a->b then b->a (nullable). */
FISR_API size_t fisr_pwc_nn_workspace_bytes(const fisr_pwc* ctx, int H, int W);
FISR_API int fisr_pwc_nn(fisr_pwc* ctx, const float* im, int H, int W, float* flow_pred, float* const* pyr, void* workspace,
                size_t workspace_bytes, void* stream);
/* the two pre/post-processing kernels on their own (parity tests): yuv [h,w,3] -> out [PH,PW,4]; flow2 [FH,FW,2] -> out [h,w,2] */
FISR_API int fisr_pwc_prep(const uint8_t* yuv, int h, int w, float* out, int PH, int PW, void* stream);
FISR_API int fisr_pwc_flow_out(const float* flow2, int FH, int FW, float* out, int h, int w, void* stream);
/* Op-level entries: one layer of the flow network exactly as the network launches it (parity tests at the sizes the
 * bench runs).  precision: FISR_PREC_F32W (float32 tensors) or FISR_PREC_F16 (fp16 feature tensors; `add` and any out_f32 /
 * in_f32 tensor stay float32 -- the flows).
 * fisr_pwc_op_conv = tf.layers.conv2d(x, cout, 3, stride, 'same', dilation_rate=dil) + leaky relu (slope; 1 =
 * linear) (+ add) of model_pwcnet.py:1092-1097, 1426-1449, 1506-1521 on the channel range [in_co, in_co + cin_buf) of a buffer
 * with pixel stride in_cs, written to the range [out_co, out_co + cout) of a buffer with pixel stride out_cs; w_host TF HWIO
 * [3,3,ci,cout]; chmap (nullable = identity): buffer channel, relative to in_co, of TF input channel j.  route 0 = the
 * network's own choice, 1 = generic implicit GEMM, 2 = persistent fp32 Winograd kernel F(2x2), 3 = FISRnet's direct kernel, 4 = its
 * fp16 LDS-DMA kernel, 5 = fp32 Winograd F(4x4) (FISR_PREC_F32W4), 6 = fp32 pointwise map to tap x output channels + 9-tap gather
 * (two output channels, linear, cin_buf % 32 == 0: the flow heads and dc_conv7), 7 = conv1a (3 -> 16 channels read from a 4-wide
 * buffer, stride 2, even sizes: one fp32 MFMA per tap) (2 - 7: error if the layer is not eligible).
 * Returns the route taken (1..7) or a negative error.
 * Synchronises the stream.
 * fisr_pwc_op_deconv = tf.layers.conv2d_transpose(x, 2, 4, 2, 'same') (:1196), w_host [4,4,2,ci];
 * fisr_pwc_op_costvol = core_costvol.cost_volume + leaky relu (:1277), 81 channels; fisr_pwc_op_warp = core_warp.dense_image_warp
 * (:1178) at (x + scale*u, y + scale*v). */
FISR_API int fisr_pwc_op_conv(const void* in, int in_cs, int in_co, int cin_buf, const float* w_host, const float* b_host, int ci, int cout,
                     const int* chmap, void* out, int out_f32, int out_cs, int out_co, const float* add, int add_cs, int add_co, int n, int h,
                     int w, int stride, int dil, float slope, int route, int precision, void* stream);
FISR_API int fisr_pwc_op_deconv(const void* in, int in_f32, int in_cs, int in_co, int cin4, const float* w_host, const float* b_host, int ci,
                       const int* chmap, void* out, int out_cs, int out_co, int n, int h, int w, int precision, void* stream);
FISR_API int fisr_pwc_op_costvol(const void* c1, const void* c2, int c, void* out, int out_cs, int out_co, int n, int h, int w, int precision, void* stream);
FISR_API int fisr_pwc_op_warp(const void* img, int c, const void* flow, int f_cs, int f_co, float scale, void* out, int n, int h, int w, int precision,
                     void* stream);

/* ---- training graph: the ops the reference gets from TensorFlow's autodiff and optimizer ----
 * FISRnet.build_model (FISRnet.py:175-497) builds forward passes, seven loss terms and tf.train.AdamOptimizer in Python;
 * tf.gradients supplies the backward ops.  fisr_amd/train.py keeps that structure (a Python tape over these entry
 * points); each entry is one HIP kernel launch on device pointers.  All tensors fp32 NHWC, weights TF HWIO.
 * In reference terms: fisr_train_conv3x3 = ops.py:7-11 Conv2d (+ fused relu / residual / depth_to_space) and, with
 * weights packed `transpose`d, its Conv2DBackpropInput; fisr_train_wgrad / _bgrad = Conv2DBackpropFilter / BiasAddGrad;
 * _relu_bwd = ReluGrad; _maxpool2_bwd = MaxPoolGrad (ops.py:54); _upsample2_bwd = ResizeBilinearGrad (ops.py:69);
 * _s2d = the gradient of tf.depth_to_space (FISRnet.py:99); _copy_channels = the gradients of tf.concat / tf.split /
 * tf.slice; _loss = FISRnet.py:316-484 for one level (values and gradients); _adam = tf.train.AdamOptimizer
 * (FISRnet.py:490-491). */
FISR_API size_t fisr_train_packed_bytes(int ci, int co, int transpose);
FISR_API int fisr_train_pack(const float* d_w_hwio, int ci, int co, int transpose, void* d_packed, void* stream);
FISR_API size_t fisr_train_wino_bytes(int ci, int co, int transpose);      /* 0: not eligible for the Winograd kernel */
FISR_API int fisr_train_pack_wino(const float* d_w_hwio, int ci, int co, int transpose, void* d_packed, void* stream);
/* every layout of n convs in ONE launch (what a training step does after Adam): d_descs is a DEVICE array of n descriptors;
 * a null destination skips that layout */
typedef struct fisr_train_pack_desc {
  const float* w;            /* HWIO [3][3][ci][co], device */
  float* pk; float* pk_t;    /* fisr_train_pack, transpose = 0 / 1 */
  void* pkw; void* pkw_t;    /* fisr_train_pack_wino, transpose = 0 / 1 */
  int ci, co;
} fisr_train_pack_desc;
FISR_API int fisr_train_pack_all(const fisr_train_pack_desc* d_descs, int n, void* stream);
/* d_packed_wino (nullable): the slabs of fisr_train_pack_wino; the call then runs the Winograd kernel and d_packed may be
 * NULL.  d_bias must be readable up to the N block's padding (cout rounded up to 64; 16 for the heads).
 * fisr_train_wgrad picks its kernel by shape: the Winograd-domain weight gradient where a 4x32 / 8x16 / 16x8 / 8x8-pixel
 * tile covers the map to >= 75 %, the direct pixel-axis GEMM otherwise, a vector-ALU kernel for co <= 8 (the heads). */
FISR_API int fisr_train_conv3x3(const float* in0, int c0, const float* in1, int c1, const void* d_packed /* nullable, see above */,
                       const float* d_bias, int cout, const float* res, float* out, int n, int h, int w, int flags,
                       int out_cstride, int out_coff, int out_split, int out_gap, const void* d_packed_wino /* nullable */,
                       void* stream);
FISR_API int fisr_train_wgrad(const float* x0, int c0, const float* x1, int c1, int relu_in, const float* g, int cg, float* dw,
                     float* db /* nullable */, int ci, int co, int n, int h, int w, void* stream);
FISR_API int fisr_train_bgrad(const float* g, int cg, size_t npix, float* db, int co, void* stream);
FISR_API int fisr_train_relu_bwd(const float* g_in, const float* ref, float* g_out, size_t count, void* stream);
FISR_API int fisr_train_axpy(const float* x, float a, float* y, size_t count, void* stream);
FISR_API int fisr_train_maxpool2_bwd(const float* x, const float* dpool, float* dx, int n, int h, int w, int c, void* stream);
FISR_API int fisr_train_upsample2_bwd(const float* dy, float* dx, int n, int h, int w, int c, void* stream);
FISR_API int fisr_train_s2d(const float* g, float* out, int n, int h, int w, int c, void* stream);
FISR_API int fisr_train_copy_channels(const float* src, int scs, int sco, float* dst, int dcs, int dco, int nc, size_t npix,
                             int add, void* stream);
FISR_API int fisr_train_loss(const float* const* pred4, const float* gt, float* const* grad4, float* sums, size_t npix,
                    const float* k7, void* stream);
FISR_API int fisr_train_adam(float* w, const float* g, float* m, float* v, size_t count, float lr_t, float b1, float b2,
                    float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FISR_H */
