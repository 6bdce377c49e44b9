// C-ABI of the on-GPU optical flow (included at the end of fisr_api.hip): PWC-Net-large as the reference's
// `FISR_for_video_Compute_Flow` runs it (FISR_tfoptflow/FISR_for_video_pwcnet_predict_from_img_test.py:84-147 over
// model_pwcnet.py:1525-1593).  Weights are installed by TF variable name ('pwcnet/featpyr/conv1a/kernel', ...), exactly
// what a `pwcnet.ckpt-595000` bundle holds (script :31); the forward is a schedule of the kernels in pwc_kernels.h.
#include "pwc_kernels.h"

namespace {

constexpr int PWC_LVLS = 6, PWC_PRED = 2;
const int PWC_CH[7] = {4, 16, 32, 64, 96, 128, 196};     // [0] = padded RGB input
const int PWC_DENSE[5] = {128, 128, 96, 64, 32};
const int PWC_CTXT[7][2] = {{128, 1}, {128, 2}, {128, 4}, {96, 8}, {64, 16}, {32, 1}, {2, 1}};

inline int pad4(int c) { return (c + 3) & ~3; }

// Channel layout of the decoder buffer D[lvl] (tf.concat order of model_pwcnet.py:1424, 1428-1445, every group padded
// to a multiple of 4): [act4 32 | act3 64 | act2 96 | act1 128 | act0 128 | corr 81(88; 96 at the top level) | c1 C | up_flow 2(4) | up_feat 2(4)]
struct DecLayout {
  int c1;                        // feature channels of the level (0 at the top level: x = corr only)
  int off_act[5];                // act0 .. act4
  int off_corr, off_c1, off_upflow, off_upfeat, total;
  explicit DecLayout(int lvl) {
    c1 = lvl == PWC_LVLS ? 0 : PWC_CH[lvl];
    off_act[4] = 0; off_act[3] = 32; off_act[2] = 96; off_act[1] = 192; off_act[0] = 320;
    // 81 cost-volume channels in a block of 88 (96 at the top level, which has nothing behind it): with the 4 + 4 of
    // the up-sampled flow / features every conv of the level then reads a channel range that is a multiple of 32
    off_corr = 448; off_c1 = off_corr + (c1 ? 88 : 96);
    off_upflow = off_c1 + c1; off_upfeat = off_upflow + (c1 ? 4 : 0);
    total = off_upfeat + (c1 ? 4 : 0);
  }
  // buffer channel (absolute) of TF channel j of the tensor that STARTS at dense stage `from` (5 = x itself,
  // i = input of conv{lvl}_i is stage 5 - ... see map_from)
  std::vector<int> map_from(int first_act /* 0..4: the newest act included, 5: none */) const {
    std::vector<int> m;
    for (int a = first_act == 5 ? -1 : first_act; a >= 0; --a)
      for (int k = 0; k < PWC_DENSE[a]; ++k) m.push_back(off_act[a] + k);
    for (int k = 0; k < 81; ++k) m.push_back(off_corr + k);
    for (int k = 0; k < c1; ++k) m.push_back(off_c1 + k);
    if (c1) { m.push_back(off_upflow); m.push_back(off_upflow + 1); m.push_back(off_upfeat); m.push_back(off_upfeat + 1); }
    return m;
  }
};

struct PwcVar {
  std::vector<int64_t> shape;
  std::vector<float> v;
  bool have = false;
};

struct PwcConv {                 // one packed convolution
  float* d_w = nullptr; float* d_b = nullptr;     // generic implicit-GEMM kernel (fp32 weights, any stride / dilation)
  Conv1aWeights* w1a = nullptr;  // conv1a (3 -> 16, stride 2): [9][4][16] + bias [16] for the vector-ALU kernel (host copy, passed by value)
  char* d_wu = nullptr;          // fp32 engine: Winograd slabs for conv3x3_wino8p_kernel (stride 1, Cout >= 32)
  char* d_wu4 = nullptr;         // fp32 engine with F(4x4) (FISR_PREC_F32W4): slabs for conv3x3_wf4_kernel<GENERAL> beside them
  void* d_wd = nullptr;          // fp16 engine: weight slabs of the LDS-DMA kernel conv3x3_dma.h (stride 1, Cout >= 16)
  void* d_wg16 = nullptr;        // fp16 engine: fp16 weights of pwc_convg_f16_kernel (the stride-2 pyramid convolutions, level 6's 196-channel layers)
  int cout_pad_d = 0, nt_d = 2;  // ... its Cout padding and N block (32 * nt_d channels: 32 when Cout % 64 == 32)
  ConvW dw;                      // FISRnet's direct kernel in the engine's arithmetic (stride 1, dilation 1: the 2-channel flow heads; fp32: also level 1)
  bool have_dw = false;
  float* d_wp = nullptr;         // fp32 engines, two output channels, cin_buf % 32 == 0: pack_pointwise() of [cin_buf][tap * 2 + o] for pwc_pointwise_f32_kernel<20>
  int cin_buf = 0, cout = 0, cout_pad = 0;
};
struct PwcDeconv {
  float* d_w = nullptr; float* d_b = nullptr; int cin4 = 0;
  void* d_wd = nullptr; float* d_bz = nullptr;   // fp16 engine, wide inputs: the 16 taps x 2 outputs as a 32-channel centre-tap conv on the LDS-DMA kernel
  float* d_wp = nullptr;                         // fp32 engines, cin4 % 32 == 0: pack_pointwise() of [cin4][32] for pwc_pointwise_f32_kernel<32> (the same 32 channels)
};

}  // namespace

struct fisr_pwc {
  int dev = 0;
  bool finalized = false;
  int precision = FISR_PREC_F32W;  // FISR_PREC_F32W: fp32 tensors and arithmetic (Winograd for the dense layers); FISR_PREC_F16: fp16 features
  bool f4 = false;                 // FISR_PREC_F32W4: the fp32 engine with F(4x4,3x3) on the maps where it is the faster Winograd kernel
  std::map<std::string, PwcVar> vars;
  std::map<std::string, PwcConv> convs;
  std::map<std::string, PwcDeconv> deconvs;
  std::string err;
};

namespace {

int pfail(fisr_pwc* c, int code, const std::string& msg) { g_err = msg; if (c) c->err = msg; return code; }

std::vector<std::pair<std::string, std::vector<int64_t>>> pwc_variable_list() {
  std::vector<std::pair<std::string, std::vector<int64_t>>> out;
  auto conv = [&](const std::string& n, int ci, int co) {
    out.push_back({n + "/kernel", {3, 3, ci, co}});
    out.push_back({n + "/bias", {co}});
  };
  const int real[7] = {3, 16, 32, 64, 96, 128, 196};
  for (int l = 1; l <= PWC_LVLS; ++l) {
    const std::string p = "pwcnet/featpyr/conv" + std::to_string(l);
    conv(p + "a", real[l - 1], real[l]); conv(p + "aa", real[l], real[l]); conv(p + "b", real[l], real[l]);
  }
  for (int l = PWC_LVLS; l >= PWC_PRED; --l) {
    int c = 81 + (l == PWC_LVLS ? 0 : real[l] + 4);
    for (int i = 0; i < 5; ++i) { conv("pwcnet/predict_flow/conv" + std::to_string(l) + "_" + std::to_string(i), c, PWC_DENSE[i]); c += PWC_DENSE[i]; }
    conv("pwcnet/predict_flow/flow" + std::to_string(l), c, 2);
    int ci = c;
    for (int i = 0; i < 7; ++i) { conv("pwcnet/ctxt/dc_conv" + std::to_string(l) + std::to_string(i + 1), ci, PWC_CTXT[i][0]); ci = PWC_CTXT[i][0]; }
    if (l != PWC_PRED) {
      out.push_back({"pwcnet/upsample/up_flow" + std::to_string(l) + "/kernel", {4, 4, 2, 2}});
      out.push_back({"pwcnet/upsample/up_flow" + std::to_string(l) + "/bias", {2}});
      out.push_back({"pwcnet/upsample/up_feat" + std::to_string(l) + "/kernel", {4, 4, 2, c}});
      out.push_back({"pwcnet/upsample/up_feat" + std::to_string(l) + "/bias", {2}});
    }
  }
  return out;
}

// pack HWIO [3,3,ci,co] for pwc_convg_kernel: [cin_buf8/8][CoutPad/64][9][64 rows][8 floats], LDS image; chmap[j] =
// buffer channel (relative to the conv's first input channel) of TF input channel j
int pwc_pack_conv(fisr_pwc* ctx, const std::string& name, const std::vector<int>& chmap, int cin_buf, PwcConv& pc, bool wino = false) {
  const PwcVar& kw = ctx->vars[name + "/kernel"];
  const PwcVar& kb = ctx->vars[name + "/bias"];
  const int ci = (int)kw.shape[2], co = (int)kw.shape[3];
  if ((int)chmap.size() != ci) return pfail(ctx, FISR_EINVAL, name + ": channel map size mismatch");
  pc.cin_buf = cin_buf; pc.cout = co; pc.cout_pad = round_up(co, G_BN);
  const int nch = (cin_buf + G_CH - 1) / G_CH, nb = pc.cout_pad / G_BN;
  std::vector<float> wp((size_t)nch * nb * 9 * G_BN * 8, 0.f), bp(pc.cout_pad, 0.f);
  for (int tap = 0; tap < 9; ++tap)
    for (int j = 0; j < ci; ++j) {
      const int c = chmap[j], kc = c / G_CH, cc = c % G_CH, h = cc >> 2, e = cc & 3;
      for (int n = 0; n < co; ++n) {
        const int blk = n / G_BN, nl = n % G_BN, wi = nl & 31, wk = wi >> 4, wr = wi & 15;
        const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
        wp[((((size_t)kc * nb + blk) * 9 + tap) * G_BN + row) * 8 + ((h ^ ((row >> 3) & 1)) * 4) + e] =
            kw.v[((size_t)tap * ci + j) * co + n];
      }
    }
  for (int n = 0; n < co; ++n) bp[n] = kb.v[n];
  HIP_OK(nullptr, hipMalloc((void**)&pc.d_w, wp.size() * 4));
  HIP_OK(nullptr, hipMalloc((void**)&pc.d_b, bp.size() * 4));
  HIP_OK(nullptr, hipMemcpy(pc.d_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(nullptr, hipMemcpy(pc.d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  if (ci == 3 && co == 16 && cin_buf == 4) {          // conv1a: [9][4][16] + bias for pwc_conv1a_kernel
    delete pc.w1a;
    pc.w1a = new Conv1aWeights();
    memset(pc.w1a, 0, sizeof(Conv1aWeights));
    for (int tap = 0; tap < 9; ++tap)
      for (int j = 0; j < 3; ++j)
        for (int n = 0; n < 16; ++n) pc.w1a->w[(size_t)tap * 64 + chmap[j] * 16 + n] = kw.v[((size_t)tap * 3 + j) * 16 + n];
    for (int n = 0; n < 16; ++n) pc.w1a->bias[n] = kb.v[n];
  }
  if (ctx->precision == FISR_PREC_F16 && cin_buf % 4 == 0 && cin_buf >= 16 && co >= 16 && (!wino || co % 16 != 0 || cin_buf % D_CH != 0)) {
    // (the layers of the fp16 engine that get no LDS-DMA slabs below: the stride-2 pyramid convolutions and level 6's 196-channel maps)
    // fp16 engine, generic kernel in fp16 arithmetic: [cin_buf/16][CoutPad/64][9][64 rows][16 halves], pwc_convg_f16_kernel's LDS image
    const int nch16 = (cin_buf + 15) / 16;
    std::vector<_Float16> w16((size_t)nch16 * nb * 9 * G_BN * 16, (_Float16)0.f);
    for (int tap = 0; tap < 9; ++tap)
      for (int j = 0; j < ci; ++j) {
        const int c = chmap[j], kc = c / 16, cc = c % 16, h = cc >> 3, e = cc & 7;
        for (int n = 0; n < co; ++n) {
          const int blk = n / G_BN, nl = n % G_BN, wi = nl & 31, wk = wi >> 4, wr = wi & 15;
          const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
          w16[((((size_t)kc * nb + blk) * 9 + tap) * G_BN + row) * 16 + ((h ^ ((row >> 3) & 1)) * 8) + e] =
              (_Float16)kw.v[((size_t)tap * ci + j) * co + n];
        }
      }
    HIP_OK(nullptr, hipMalloc(&pc.d_wg16, w16.size() * 2));
    HIP_OK(nullptr, hipMemcpy(pc.d_wg16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
  }
  if (!wino) return 0;
  // the same kernel scattered to the buffer channels it reads, for FISRnet's fast kernels
  std::vector<float> dense((size_t)9 * cin_buf * co, 0.f);
  for (int tap = 0; tap < 9; ++tap)
    for (int j = 0; j < ci; ++j)
      for (int n = 0; n < co; ++n) dense[((size_t)tap * cin_buf + chmap[j]) * co + n] = kw.v[((size_t)tap * ci + j) * co + n];
  if (ctx->precision == FISR_PREC_F16) {
    // fp16 engine: every stride-1 layer with 16 or more output channels (any dilation) on the LDS-DMA kernel; the 2-channel flow
    // heads on FISRnet's 16-row direct kernel with fp32 output
    if (co >= 16 && co % 16 == 0 && cin_buf % D_CH == 0) {
      std::vector<char> wd;
      pc.nt_d = co % 64 == 32 ? 1 : 2;               // 32 / 96 output channels: 32-channel N blocks, nothing computed for padding
      pc.cout_pad_d = round_up(co, 32 * pc.nt_d);
      pack_weights_dma(dense.data(), cin_buf, co, cin_buf, pc.cout_pad_d, wd, 32 * pc.nt_d);
      HIP_OK(nullptr, hipMalloc(&pc.d_wd, wd.size()));
      HIP_OK(nullptr, hipMemcpy(pc.d_wd, wd.data(), wd.size(), hipMemcpyHostToDevice));
    } else if (co < 16 && cin_buf % 32 == 0) {
      pc.dw.ci = cin_buf; pc.dw.co = co; pc.dw.w = std::move(dense); pc.dw.b = kb.v;
      int rc = upload_conv<_Float16>(nullptr, pc.dw, false, false);
      if (rc) return rc;
      pc.have_dw = true;
    }
    return 0;
  }
  if (co == 2 && cin_buf % PW_CH == 0) {
    // fp32 engines: the two-channel layers as a pointwise map to tap x output channels + a 9-tap gather (pwc_kernels.h)
    std::vector<float> pw((size_t)cin_buf * 20, 0.f);
    for (int tap = 0; tap < 9; ++tap)
      for (int c = 0; c < cin_buf; ++c)
        for (int o = 0; o < 2; ++o) pw[(size_t)c * 20 + tap * 2 + o] = dense[((size_t)tap * cin_buf + c) * co + o];
    std::vector<float> pk;
    pack_pointwise(pw.data(), cin_buf, 20, pk);
    HIP_OK(nullptr, hipMalloc((void**)&pc.d_wp, pk.size() * 4));
    HIP_OK(nullptr, hipMemcpy(pc.d_wp, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
  }
  // fp32 engine: the stride-1 layers with 32 or more output channels (the dense flow estimators, the context convs incl. the
  // dilated ones: 97 % of the network's FLOPs) get FISRnet's Winograd slabs, the ones with fewer (the 2-channel flow heads, which
  // the 64-wide generic kernel computes 32 times over, and the 16-channel level-1 features) the weights of its direct kernel
  const bool as_wino = co >= 32 && co % 16 == 0 && cin_buf >= 32 && cin_buf % W_CH == 0;
  const bool as_direct = !as_wino && co < 32 && cin_buf % 16 == 0;
  if (as_wino) {
    std::vector<char> wu;
    pack_weights_wino(dense.data(), cin_buf, co, cin_buf, wu);
    HIP_OK(nullptr, hipMalloc((void**)&pc.d_wu, wu.size()));
    HIP_OK(nullptr, hipMemcpy(pc.d_wu, wu.data(), wu.size(), hipMemcpyHostToDevice));
    if (ctx->f4 && co % 4 == 0 && cin_buf % (2 * F4_CH) == 0) {
      std::vector<char> wu4;
      pack_weights_wf4(dense.data(), cin_buf, co, cin_buf, wu4);
      HIP_OK(nullptr, hipMalloc((void**)&pc.d_wu4, wu4.size()));
      HIP_OK(nullptr, hipMemcpy(pc.d_wu4, wu4.data(), wu4.size(), hipMemcpyHostToDevice));
    }
  } else if (as_direct) {
    pc.dw.ci = cin_buf; pc.dw.co = co; pc.dw.w = std::move(dense); pc.dw.b = kb.v;
    pc.dw.rows16 = true;
    int rc = upload_conv<float>(nullptr, pc.dw, false);
    if (rc) return rc;
    pc.have_dw = true;
  }
  return 0;
}

int pwc_pack_deconv(fisr_pwc* ctx, const std::string& name, const std::vector<int>& chmap, int cin4, PwcDeconv& pd) {
  const PwcVar& kw = ctx->vars[name + "/kernel"];   // [4,4,2,ci]
  const PwcVar& kb = ctx->vars[name + "/bias"];
  const int ci = (int)kw.shape[3];
  if ((int)chmap.size() != ci) return pfail(ctx, FISR_EINVAL, name + ": channel map size mismatch");
  pd.cin4 = cin4;
  std::vector<float> wp((size_t)16 * 2 * cin4, 0.f);
  for (int k = 0; k < 16; ++k)
    for (int o = 0; o < 2; ++o)
      for (int j = 0; j < ci; ++j) wp[((size_t)k * 2 + o) * cin4 + chmap[j]] = kw.v[((size_t)k * 2 + o) * ci + j];
  HIP_OK(nullptr, hipMalloc((void**)&pd.d_w, wp.size() * 4));
  HIP_OK(nullptr, hipMalloc((void**)&pd.d_b, 8));
  HIP_OK(nullptr, hipMemcpy(pd.d_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(nullptr, hipMemcpy(pd.d_b, kb.v.data(), 8, hipMemcpyHostToDevice));
  if (ctx->precision != FISR_PREC_F16 && cin4 % PW_CH == 0) {
    // fp32 engines: P[pixel][tap * 2 + o] by the pointwise kernel, then the same 2x2 gather
    std::vector<float> pw((size_t)cin4 * 32, 0.f);
    for (int k = 0; k < 16; ++k)
      for (int o = 0; o < 2; ++o)
        for (int j = 0; j < ci; ++j) pw[(size_t)chmap[j] * 32 + k * 2 + o] = kw.v[((size_t)k * 2 + o) * ci + j];
    std::vector<float> pk;
    pack_pointwise(pw.data(), cin4, 32, pk);
    HIP_OK(nullptr, hipMalloc((void**)&pd.d_wp, pk.size() * 4));
    HIP_OK(nullptr, hipMemcpy(pd.d_wp, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
  }
  if (ctx->precision == FISR_PREC_F16 && cin4 >= 32 && cin4 % D_CH == 0) {
    // P[pixel][tap * 2 + o] as a 3x3 convolution with 32 output channels whose only non-zero tap is the centre
    std::vector<float> dense((size_t)9 * cin4 * 32, 0.f);
    for (int k = 0; k < 16; ++k)
      for (int o = 0; o < 2; ++o)
        for (int j = 0; j < ci; ++j) dense[((size_t)4 * cin4 + chmap[j]) * 32 + k * 2 + o] = kw.v[((size_t)k * 2 + o) * ci + j];
    std::vector<char> wd;
    pack_weights_dma(dense.data(), cin4, 32, cin4, 32, wd, 32);        // one 32-channel N block (NT = 1)
    HIP_OK(nullptr, hipMalloc(&pd.d_wd, wd.size()));
    HIP_OK(nullptr, hipMemcpy(pd.d_wd, wd.data(), wd.size(), hipMemcpyHostToDevice));
    HIP_OK(nullptr, hipMalloc((void**)&pd.d_bz, D_BN * 4));
    HIP_OK(nullptr, hipMemset(pd.d_bz, 0, D_BN * 4));
  }
  return 0;
}

inline PwcItems identity_items(int n) { PwcItems it; it.n = n; for (int i = 0; i < PWC_MAX_ITEMS; ++i) it.a[i] = it.b[i] = i < n ? i : 0; return it; }
inline PwcItems swapped(const PwcItems& x) { PwcItems it = x; for (int i = 0; i < PWC_MAX_ITEMS; ++i) { it.a[i] = x.b[i]; it.b[i] = x.a[i]; } return it; }

template <typename TE>
hipError_t launch_costvol(const TE* c1, int c1_cs, int c1_co, const PwcItems& c1_img, const TE* c2, const PwcItems& c2_img, int C, TE* out,
                          int out_cs, int out_co, int n, int h, int w, hipStream_t st, int zero_pad = 0) {
  if (zero_pad && (zero_pad + 1) % 4) return hipErrorInvalidValue;        // 81 + zero_pad channels in 4-channel stores
  static bool cv_attr[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !cv_attr[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pwc_costvol_kernel<TE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)costvol_lds_bytes());
    if (e != hipSuccess) return e;
    cv_attr[dev] = true;
  }
  const int cv_tiles = ((w + TILE_W - 1) / TILE_W) * ((h + TILE_H - 1) / TILE_H) * n;
  const size_t cv_lds = std::is_same<TE, _Float16>::value ? (size_t)CV_HH * CV_HW * CV_REC16 : costvol_lds_bytes();
  hipLaunchKernelGGL(pwc_costvol_kernel<TE>, dim3(cv_tiles), dim3(256), cv_lds, st, c1, c1_cs, c1_co, c1_img, c2, c2_img, C, out,
                     out_cs, out_co, n, h, w, zero_pad);
  return hipGetLastError();
}

std::vector<int> iota_map(int n) { std::vector<int> m(n); for (int i = 0; i < n; ++i) m[i] = i; return m; }

// float32 [n] -> TE [n] (the network entry fisr_pwc_nn takes a float32 image pair)
template <typename TE>
__global__ void pwc_cast_kernel(const float* __restrict__ src, TE* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    PwcElem<TE>::st4(dst + 4 * i, reinterpret_cast<const f32x4*>(src)[i]);
}

// The flow network on feature tensors of element type TE (float: the exact engine; _Float16: FISR_PREC_F16).  Flows -- the
// outputs of the flow heads, the refined flows handed to the next level and to the caller -- are float32 in both.
template <typename TE>
struct PwcRunner {
  fisr_pwc* ctx; hipStream_t st; Arena ar; int rc = 0;
  static constexpr bool HALF = std::is_same<TE, _Float16>::value;
  TE* ealloc(size_t n) { return (TE*)ar.alloc(n * sizeof(TE)); }
  float* falloc(size_t n) { return (float*)ar.alloc(n * sizeof(float)); }
  void check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess && rc == 0) rc = pfail(ctx, FISR_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  }
  void zero(void* p, size_t bytes) { if (!ar.dry && !rc) (void)hipMemsetAsync(p, 0, bytes, st); }

  // which kernel runs a layer: 2 = FISRnet's persistent fp32 Winograd kernel (stride 1, Cout >= 32, any dilation), 4 = its fp16
  // fp32 engines: 6 = the two-channel layers without activation (flow heads, dc_conv7) as pointwise map + 9-tap gather;
  // LDS-DMA kernel (stride 1, Cout >= 16, any dilation), 3 = its direct kernel (stride 1, dilation 1: the 2-channel flow heads;
  // fp32 engine: also the 16-channel level-1 features), 1 = the generic implicit GEMM (stride 2, the residual dc_conv7)
  int conv_route(const std::string& name, int n, int h, int w, int in_cs, int out_cs, bool out_f32, int stride, int dil, float slope, bool has_add) {
    const PwcConv& pc = ctx->convs[name];
    const bool act_ok = slope == 1.f || (slope > 0.f && slope < 1.f);
    // (both engines -- 7: conv1a -- 3 (4) -> 16 channels, stride 2, even sizes: one MFMA per tap)
    if (pc.w1a && stride == 2 && dil == 1 && !has_add && in_cs == 4 && out_cs == 16 && !(h & 1) && !(w & 1) && slope != 1.f && act_ok) return 7;
    if (HALF) {
      if (pc.d_wd && stride == 1 && !has_add && !out_f32 && act_ok && dma_fits(h, w, pc.cin_buf, 0, in_cs, 0)) return 4;
      if (pc.have_dw && stride == 1 && dil == 1 && out_f32 && act_ok && pc.dw.nt == 0) return 3;     // (with or without the float32 add)
      return 1;
    }
    (void)n;
    // (6: two output channels, no activation -- pointwise map to tap x output channels + gather; reads its input once)
    if (pc.d_wp && stride == 1 && dil == 1 && slope == 1.f && in_cs % 4 == 0) return 6;
    // (5: the F(4x4) kernel, where it is the faster of the two -- the size rule of the FISRnet engine on the SUB-image of a dilated layer)
    if (pc.d_wu4 && stride == 1 && !has_add && act_ok && wf4_fits_general(h, w, pc.cin_buf, in_cs, out_cs) &&
        wf4_wins((h + dil - 1) / dil, (w + dil - 1) / dil, pc.cin_buf)) return 5;
    if (pc.d_wu && stride == 1 && !has_add && act_ok && wino_fits(1, h, w, in_cs, 0, out_cs)) return 2;
    if (pc.have_dw && stride == 1 && dil == 1 && !has_add && act_ok) return 3;
    return 1;
  }
  // out: TE (out_f32 = false) or float32 (out_f32 = true: the flow heads and dc_conv7); add: float32
  void conv(const std::string& name, const TE* in, int in_cs, int in_co, void* out, bool out_f32, int out_cs, int out_co,
            int n, int h, int w, int stride, int dil, float slope, const float* add = nullptr, int add_cs = 0, int add_co = 0) {
    if (rc) return;
    if (!HALF) out_f32 = true;
    const PwcConv& pc = ctx->convs[name];
    const int route = conv_route(name, n, h, w, in_cs, out_cs, HALF ? out_f32 : false, stride, dil, slope, add != nullptr);
    if (route == 6) {
      // T lives until the gather has run (stream order): the arena hands its bytes to the next allocation
      const size_t keep = ar.off;
      float* T = falloc((size_t)n * h * w * 20);
      ar.off = keep;
      if (ar.dry) return;
      if (in_co % 4) { rc = pfail(ctx, FISR_EINVAL, name + ": channel offset of a pointwise layer must be a multiple of 4"); return; }
      PointwiseArgs pa;
      pa.in = (const float*)in; pa.in_cs = in_cs; pa.in_co = in_co; pa.Cin = pc.cin_buf; pa.w = pc.d_wp; pa.out = T; pa.npix = (size_t)n * h * w;
      hipLaunchKernelGGL(pwc_pointwise_f32_kernel<20>, dim3((unsigned)((pa.npix + PW_PX - 1) / PW_PX)), dim3(256), 0, st, pa);
      hipLaunchKernelGGL(pwc_conv3_combine_kernel, dim3(grid_for(pa.npix)), dim3(256), 0, st, T, pc.d_b, add, add_cs, add_co, (float*)out, out_cs,
                         out_co, n, h, w);
      check(name.c_str());
      return;
    }
    if (ar.dry) return;
    if (route == 7) {
      if (in_co || out_co) { rc = pfail(ctx, FISR_EINVAL, name + ": conv1a reads and writes whole buffers"); return; }
      if constexpr (HALF)
        hipLaunchKernelGGL(pwc_conv1a_f16_kernel, dim3(grid_for((size_t)n * (h / 2) * (w / 2) * 4)), dim3(256), 0, st, in, *pc.w1a, (TE*)out, n, h, w, slope);
      else
        hipLaunchKernelGGL(pwc_conv1a_kernel<TE>, dim3(grid_for((size_t)n * (h / 2) * (w / 2) * 4)), dim3(256), 0, st, in, *pc.w1a, (TE*)out, n, h, w, slope);
      check(name.c_str());
      return;
    }
    ConvArgs a;
    a.in0 = in + in_co; a.in1 = nullptr; a.bias = pc.d_b; a.res = nullptr; a.out = out;
    a.N = n; a.H = h; a.W = w; a.Cout = pc.cout;
    a.relu_in = 0; a.relu_out = slope != 1.f; a.d2s = 0; a.d2s_shift = 0;
    a.out_cstride = out_cs; a.out_coff = out_co; a.out_split = 1 << 30; a.out_gap = 0; a.wexp = 0;
    a.in0_cs = in_cs; a.in1_cs = 0; a.slope = slope != 1.f ? slope : 0.f; a.trace = nullptr;
    if (route == 5) {
      a.wpk = pc.d_wu4;
      a.C0 = pc.cin_buf; a.C1 = 0; a.CoutPad = round_up(pc.cout, F4_BN);
      a.rec_cs = out_cs; a.rec_co = out_co; a.dil = dil;
      hipError_t e = launch_conv_wf4(a, st);
      if (e != hipSuccess && rc == 0) rc = pfail(ctx, FISR_EHIP, name + " (winograd F(4x4)): " + hipGetErrorString(e));
      return;
    }
    if (route == 2 || route == 4) {
      // FISRnet's persistent Winograd kernel / LDS-DMA kernel on a channel range of the buffer
      a.wpk = route == 2 ? (const void*)pc.d_wu : pc.d_wd;
      a.C0 = pc.cin_buf; a.C1 = 0; a.CoutPad = route == 2 ? round_up(pc.cout, W_BN) : pc.cout_pad_d;
      a.rec_cs = out_cs; a.rec_co = out_co; a.dil = dil;
      hipError_t e = hipSuccess;
      if (route == 4) e = launch_conv_dma(a, st, pc.nt_d);
      else if (wino_fits(n, h, w, in_cs, 0, out_cs)) e = launch_conv_wino(a, st);
      else {
        // the Winograd kernel addresses its whole input batch with 32-bit byte offsets: one image per launch when the batch is too big
        a.N = 1;
        for (int k = 0; k < n && e == hipSuccess; ++k) {
          a.in0 = in + in_co + (size_t)k * h * w * in_cs;
          a.out = (float*)out + (size_t)k * h * w * out_cs;
          e = launch_conv_wino(a, st);
        }
      }
      if (e != hipSuccess && rc == 0) rc = pfail(ctx, FISR_EHIP, name + (route == 2 ? " (winograd): " : " (lds-dma): ") + hipGetErrorString(e));
      return;
    }
    if (route == 3) {
      if (add && (add_cs != out_cs || add_co != out_co)) { if (rc == 0) rc = pfail(ctx, FISR_EINVAL, name + ": the added tensor must have the output's layout"); return; }
      a.res = add;                                                   // (16-row kernel, fp32 store: added after the activation)
      a.wpk = pc.dw.d_w; a.bias = pc.dw.d_b;
      a.C0 = pc.dw.cin_pad; a.C1 = 0; a.CoutPad = pc.dw.cout_pad;
      a.rec_cs = pc.cout; a.rec_co = 0; a.dil = 1;
      const bool scatter = HALF || !(out_cs == pc.cout && out_co == 0);     // dense records, or per-channel fp32 stores
      hipError_t e = launch_conv<TE>(a, pc.dw.nt, scatter, st);
      if (e != hipSuccess && rc == 0) rc = pfail(ctx, FISR_EHIP, name + " (direct): " + hipGetErrorString(e));
      return;
    }
    static bool attr_done[64] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwc_convg_kernel<TE, TE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)convg_lds_bytes());
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwc_convg_kernel<TE, float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)convg_lds_bytes());
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    ConvGArgs g;
    g.in = in; g.in_cs = in_cs; g.in_co = in_co; g.Cin = pc.cin_buf; g.w = pc.d_w; g.bias = pc.d_b;
    g.out = out; g.out_cs = out_cs; g.out_co = out_co; g.Cout = pc.cout; g.CoutPad = pc.cout_pad;
    g.add = add; g.add_cs = add_cs; g.add_co = add_co;
    g.N = n; g.H = h; g.W = w; g.stride = stride; g.dil = dil; g.slope = slope;
    g.OH = (h + stride - 1) / stride; g.OW = (w + stride - 1) / stride;
    const int tot_h = std::max((g.OH - 1) * stride + 2 * dil + 1 - h, 0), tot_w = std::max((g.OW - 1) * stride + 2 * dil + 1 - w, 0);
    g.pad_t = tot_h / 2; g.pad_l = tot_w / 2;                      // TF 'SAME': the smaller half goes first
    const int tiles = ((g.OW + TILE_W - 1) / TILE_W) * ((g.OH + TILE_H - 1) / TILE_H) * n;
    if (HALF && pc.d_wg16 && !out_f32 && !add && in_cs % 4 == 0 && in_co % 4 == 0) {
      // fp16 engine: the generic kernel on the fp16 matrix pipe (the stride-2 pyramid convolutions)
      static bool attr16_done[64] = {};
      if (dev < 0 || dev >= 64 || !attr16_done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwc_convg_f16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)convg16_lds_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwc_convg_f16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)convg16_lds_bytes());
        if (dev >= 0 && dev < 64) attr16_done[dev] = true;
      }
      g.w = (const float*)pc.d_wg16;
      const int tiles16 = ((g.OW + 31) / 32) * ((g.OH + G16_TH - 1) / G16_TH) * n;
      if (pc.cin_buf % 16 == 0 && in_cs % 8 == 0 && in_co % 8 == 0)
        hipLaunchKernelGGL(pwc_convg_f16_kernel<true>, dim3(tiles16 * (pc.cout_pad / G_BN)), dim3(256), convg16_lds_bytes(), st, g);
      else
        hipLaunchKernelGGL(pwc_convg_f16_kernel<false>, dim3(tiles16 * (pc.cout_pad / G_BN)), dim3(256), convg16_lds_bytes(), st, g);
      check(name.c_str());
      return;
    }
    if (out_f32) hipLaunchKernelGGL((pwc_convg_kernel<TE, float>), dim3(tiles * (pc.cout_pad / G_BN)), dim3(256), convg_lds_bytes(), st, g);
    else hipLaunchKernelGGL((pwc_convg_kernel<TE, TE>), dim3(tiles * (pc.cout_pad / G_BN)), dim3(256), convg_lds_bytes(), st, g);
    check(name.c_str());
  }
  template <typename TI>
  void deconv(const std::string& name, const TI* in, int in_cs, int in_co, TE* out, int out_cs, int out_co, int n, int h, int w, bool pad4 = false) {
    const int p4 = pad4 && out_cs % 4 == 0 && out_co % 4 == 0 ? 1 : 0;     // (pad4: the decoder's call -- channels out_co + 2, + 3 are padding, written as zeros)
    if (pad4 && !p4) { if (rc == 0) rc = pfail(ctx, FISR_EINVAL, name + ": padded pair needs a 4-channel aligned slot"); return; }
    const PwcDeconv& pd = ctx->deconvs[name];
    if (HALF && std::is_same<TI, TE>::value && pd.d_wd && dma_fits(h, w, pd.cin4, 0, in_cs, 0)) {
      // wide input: taps x outputs as 32 channels of a centre-tap convolution on the matrix pipe, then the 2x2 gather
      TE* P = ealloc((size_t)n * h * w * 32);
      if (rc || ar.dry) return;
      ConvArgs a;
      a.in0 = (const TE*)in + in_co; a.in1 = nullptr; a.wpk = pd.d_wd; a.bias = pd.d_bz; a.res = nullptr; a.out = P;
      a.C0 = pd.cin4; a.C1 = 0; a.N = n; a.H = h; a.W = w; a.Cout = 32; a.CoutPad = 32;
      a.relu_in = 0; a.relu_out = 0; a.d2s = 0; a.d2s_shift = 0;
      a.out_cstride = 32; a.out_coff = 0; a.out_split = 1 << 30; a.out_gap = 0; a.wexp = 0;
      a.in0_cs = in_cs; a.in1_cs = 0; a.rec_cs = 32; a.rec_co = 0; a.slope = 0.f; a.dil = 1; a.trace = nullptr;
      hipError_t e = launch_conv_dma(a, st, 1);
      if (e != hipSuccess && rc == 0) { rc = pfail(ctx, FISR_EHIP, name + " (lds-dma): " + hipGetErrorString(e)); return; }
      hipLaunchKernelGGL(pwc_deconv_combine_kernel<TE>, dim3(grid_for((size_t)n * 4 * h * w)), dim3(256), 0, st, P, pd.d_b, out, out_cs, out_co, n, h, w, p4);
      check(name.c_str());
      return;
    }
    if (!HALF && std::is_same<TI, float>::value && pd.d_wp && in_cs % 4 == 0 && in_co % 4 == 0) {
      // wide input, fp32 engines: the 16 taps x 2 outputs as a pointwise map (the input is read once), then the 2x2 gather
      const size_t keep = ar.off;
      float* P = falloc((size_t)n * h * w * 32);
      ar.off = keep;
      if (rc || ar.dry) return;
      PointwiseArgs pa;
      pa.in = (const float*)in; pa.in_cs = in_cs; pa.in_co = in_co; pa.Cin = pd.cin4; pa.w = pd.d_wp; pa.out = P; pa.npix = (size_t)n * h * w;
      hipLaunchKernelGGL(pwc_pointwise_f32_kernel<32>, dim3((unsigned)((pa.npix + PW_PX - 1) / PW_PX)), dim3(256), 0, st, pa);
      hipLaunchKernelGGL((pwc_deconv_combine_kernel<float, true>), dim3(grid_for((size_t)n * 4 * h * w)), dim3(256), 0, st, P, pd.d_b, (float*)out, out_cs,
                         out_co, n, h, w, p4);
      check(name.c_str());
      return;
    }
    if (rc || ar.dry) return;
    if (pd.cin4 == 4) {
      hipLaunchKernelGGL((pwc_deconv4_kernel<TI, TE>), dim3(grid_for((size_t)n * 4 * h * w)), dim3(256), 0, st, in, in_cs, in_co, pd.d_w, pd.d_b, out,
                         out_cs, out_co, n, h, w, p4);
      check(name.c_str());
      return;
    }
    hipLaunchKernelGGL((pwc_deconv_kernel<TI, TE>), dim3(grid_for((size_t)n * 4 * h * w * 8)), dim3(256), 0, st, in, in_cs, in_co, pd.cin4,
                       pd.d_w, pd.d_b, out, out_cs, out_co, n, h, w, p4);
    check(name.c_str());
  }

  int hh[7], ww[7];
  TE* F[7];

  // extract_features (model_pwcnet.py:1012-1101) of nf prepared frames at once: im [nf, H, W, 4] (RGB / 255 + a zero channel;
  // H, W multiples of 64) -> F[1..6] [nf, H/2^l, W/2^l, C_l]
  void pyramid(const TE* im, int nf, int H, int W) {
    F[0] = const_cast<TE*>(im);
    hh[0] = H; ww[0] = W;
    for (int l = 1; l <= PWC_LVLS; ++l) {
      hh[l] = hh[l - 1] / 2; ww[l] = ww[l - 1] / 2;
      const size_t px = (size_t)nf * hh[l] * ww[l];
      const size_t mark = ar.off;
      F[l] = ealloc(px * PWC_CH[l]);
      const size_t keep = ar.off;
      TE* A = ealloc(px * PWC_CH[l]); TE* B = ealloc(px * PWC_CH[l]);
      const std::string p = "pwcnet/featpyr/conv" + std::to_string(l);
      conv(p + "a", F[l - 1], PWC_CH[l - 1], 0, A, false, PWC_CH[l], 0, nf, hh[l - 1], ww[l - 1], 2, 1, 0.1f);     // (l == 1: route 7)
      conv(p + "aa", A, PWC_CH[l], 0, B, false, PWC_CH[l], 0, nf, hh[l], ww[l], 1, 1, 0.1f);
      conv(p + "b", B, PWC_CH[l], 0, F[l], false, PWC_CH[l], 0, nf, hh[l], ww[l], 1, 1, 0.1f);
      (void)mark;
      ar.off = keep;                                       // the two temporaries of a level are dead once its features exist
    }
  }

  // The coarse-to-fine decoder (model_pwcnet.py:1546-1593) for items.n (frame a, frame b) items at once (batch axis = item).
  // flow2: refined level-2 flows [items.n, H/4, W/4] x stride 4 floats.  pyr_out (nullable): 5 pointers per item, the refined
  // flows of levels 6..2 as dense [h_l, w_l, 2].
  int decode(const PwcItems& items, float** flow2, float* const* pyr_out) {
    const int N = items.n;
    const PwcItems ident = identity_items(N), items_b = swapped(items);
    TE* Dprev = nullptr; int prev_total = 0;
    float* flow_prev = nullptr;
    for (int l = PWC_LVLS; l >= PWC_PRED; --l) {
      const DecLayout L(l);
      const int h = hh[l], w = ww[l];
      const size_t px = (size_t)h * w, npx = px * N;
      TE* D = ealloc(npx * L.total);
      // channel padding must read as finite zeros (its weights are zero): written by the kernels that fill the group in front of it
      const int corr_pad = L.off_c1 - (L.off_corr + 81);          // 7 (15 at the top level): written by the cost volume kernel
      // (the padding behind up_flow / up_feat is written by their transpose convolutions: deconv(..., pad4))
      const std::string ls = std::to_string(l);
      if (l != PWC_LVLS) {
        // up-sampled flow / features of the level above land directly in this level's buffer (:1577-1578, :1424)
        deconv<float>("pwcnet/upsample/up_flow" + std::to_string(l + 1), flow_prev, 4, 0, D, L.total, L.off_upflow, N, hh[l + 1], ww[l + 1], true);
        deconv<TE>("pwcnet/upsample/up_feat" + std::to_string(l + 1), Dprev, prev_total, 0, D, L.total, L.off_upfeat, N, hh[l + 1], ww[l + 1], true);
        TE* Wp = ealloc(npx * PWC_CH[l]);
        if (!rc && !ar.dry) {
          hipLaunchKernelGGL(pwc_warp_kernel<TE>, dim3(grid_for(npx * PWC_CH[l] / 4)), dim3(256), 0, st, F[l], items_b, PWC_CH[l], D, L.total,
                             L.off_upflow, 20.f / (float)(1 << l), Wp, N, h, w);      // :1560-1561: warp the features of frame b
          check("warp");
          hipLaunchKernelGGL(pwc_copy_channels_kernel<TE>, dim3(grid_for(npx * PWC_CH[l] / 4)), dim3(256), 0, st, F[l], items, PWC_CH[l], D,
                             L.total, L.off_c1, px, N);
          check("copy c1");
          // (c1 read from the level's own dense tensor, not from its 1152-byte-stride copy in the buffer)
          hipError_t e = launch_costvol<TE>(F[l], PWC_CH[l], 0, items, Wp, ident, PWC_CH[l], D, L.total, L.off_corr, N, h, w, st, corr_pad);   // :1277
          if (e != hipSuccess && rc == 0) rc = pfail(ctx, FISR_EHIP, std::string("cost volume: ") + hipGetErrorString(e));
        }
      } else if (!rc && !ar.dry) {
        hipError_t e = launch_costvol<TE>(F[l], PWC_CH[l], 0, items, F[l], items_b, PWC_CH[l], D, L.total, L.off_corr, N, h, w, st, corr_pad);
        if (e != hipSuccess && rc == 0) rc = pfail(ctx, FISR_EHIP, std::string("cost volume: ") + hipGetErrorString(e));
      }
      for (int i = 0; i < 5; ++i)                                  // predict_flow :1426-1445 (dense connections)
        conv("pwcnet/predict_flow/conv" + ls + "_" + std::to_string(i), D, L.total, i == 0 ? L.off_corr : L.off_act[i - 1],
             D, false, L.total, L.off_act[i], N, h, w, 1, 1, 0.1f);
      float* flow = falloc(npx * 4); zero(flow, npx * 4 * sizeof(float));
      conv("pwcnet/predict_flow/flow" + ls, D, L.total, 0, flow, true, 4, 0, N, h, w, 1, 1, 1.f);           // :1447
      TE* T1 = ealloc(npx * 128); TE* T2 = ealloc(npx * 128);
      const TE* src = D; int src_cs = L.total;                     // refine_flow :1506-1521
      for (int i = 0; i < 7; ++i) {
        const std::string cn = "pwcnet/ctxt/dc_conv" + ls + std::to_string(i + 1);
        TE* dst = (i & 1) ? T2 : T1;
        if (i < 6) {
          conv(cn, src, src_cs, 0, dst, false, PWC_CTXT[i][0], 0, N, h, w, 1, PWC_CTXT[i][1], 0.1f);
          src = dst; src_cs = PWC_CTXT[i][0];
        } else {
          float* refined = falloc(npx * 4); zero(refined, npx * 4 * sizeof(float));
          conv(cn, src, src_cs, 0, refined, true, 4, 0, N, h, w, 1, 1, 1.f, flow, 4, 0);
          flow_prev = refined;
        }
      }
      if (pyr_out && !rc && !ar.dry)
        for (int k = 0; k < N; ++k)
          if (pyr_out[k * 5 + (PWC_LVLS - l)])
            hipLaunchKernelGGL(stitch_free_copy2_kernel, dim3(grid_for(px)), dim3(256), 0, st, flow_prev + (size_t)k * px * 4,
                               pyr_out[k * 5 + (PWC_LVLS - l)], px);
      Dprev = D; prev_total = L.total;
    }
    *flow2 = flow_prev;
    return rc;
  }
};

template <typename F>
auto with_pwc_elem(const fisr_pwc* c, F&& f) {
  if (c->precision == FISR_PREC_F16) return f(_Float16());
  return f(float());
}

// One call of the flow network on a set of frames: prepared images (im32 float32 [nf, H, W, 4], or NULL and yuv frames [h, w, 3]
// that the pre-processing kernel turns into them), the feature pyramid of every frame ONCE, then the decoder over the items
// in groups of PWC_MAX_ITEMS.  each(k, flow2_k): called per item with its refined level-2 flow ([H/4, W/4] x stride 4 floats).
template <typename TE, typename Each>
int pwc_run(fisr_pwc* c, const float* im32, const uint8_t* const* yuv, int nf, int h, int w, int H, int W, const std::vector<std::pair<int, int>>& pairs,
            float* const* pyr, void* ws, size_t ws_bytes, hipStream_t st, bool dry, size_t* peak, Each&& each) {
  static const ColorConsts cc = make_color_consts();
  PwcRunner<TE> r; r.ctx = c; r.st = st; r.ar.base = (char*)ws; r.ar.cap = ws_bytes; r.ar.dry = dry;
  TE* im = nullptr;
  if (std::is_same<TE, float>::value && im32) im = (TE*)const_cast<float*>(im32);
  else {
    im = r.ealloc((size_t)nf * H * W * 4);
    if (!dry) {
      if (im32) hipLaunchKernelGGL(pwc_cast_kernel<TE>, dim3(grid_for((size_t)nf * H * W)), dim3(256), 0, st, im32, im, (size_t)nf * H * W);
      else
        for (int k = 0; k < nf; ++k)
          hipLaunchKernelGGL(pwc_prep_kernel<TE>, dim3(grid_for((size_t)H * W)), dim3(256), 0, st, yuv[k], h, w, im + (size_t)k * H * W * 4, H, W, cc);
      HIP_OK(nullptr, hipGetLastError());
    }
  }
  r.pyramid(im, nf, H, W);
  const size_t mark = r.ar.off;
  for (size_t g = 0; g < pairs.size() && !r.rc; g += PWC_MAX_ITEMS) {
    PwcItems items = identity_items(0);
    items.n = (int)std::min<size_t>(PWC_MAX_ITEMS, pairs.size() - g);
    for (int k = 0; k < items.n; ++k) { items.a[k] = pairs[g + k].first; items.b[k] = pairs[g + k].second; }
    r.ar.off = mark;
    float* f2 = nullptr;
    int rc = r.decode(items, &f2, pyr ? pyr + g * 5 : nullptr);
    if (rc) return rc;
    if (!dry)
      for (int k = 0; k < items.n; ++k) {
        rc = each((int)g + k, f2 + (size_t)k * (H / 4) * (W / 4) * 4);
        if (rc) return rc;
      }
  }
  if (peak) *peak = r.ar.peak + 256;
  return r.rc;
}

size_t pwc_ws(fisr_pwc* c, int nf, int H, int W, int npairs, bool with_prep) {
  std::vector<std::pair<int, int>> pairs;
  for (int k = 0; k < npairs; ++k) pairs.push_back({0, nf > 1 ? 1 : 0});
  size_t peak = 0;
  with_pwc_elem(c, [&](auto tag) {
    typedef decltype(tag) TE;
    const float* fake32 = with_prep ? nullptr : (const float*)16;
    return pwc_run<TE>(c, fake32, nullptr, nf, H / 2, W / 2, H, W, pairs, nullptr, nullptr, 0, nullptr, true, &peak, [](int, float*) { return 0; });
  });
  return peak;
}

}  // namespace

extern "C" {

int fisr_pwc_create(fisr_pwc** out, int device_id) {
  if (!out) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_create: out is NULL");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
    return pfail(nullptr, FISR_EHIP, "fisr_pwc_create: no such HIP device");
  fisr_pwc* c = new fisr_pwc();
  c->dev = device_id;
  for (auto& v : pwc_variable_list()) c->vars[v.first].shape = v.second;
  *out = c;
  return 0;
}

static void pwc_release_packed(fisr_pwc* c) {
  for (auto& kv : c->convs) {
    PwcConv& pc = kv.second;
    if (pc.d_w) (void)hipFree(pc.d_w); if (pc.d_b) (void)hipFree(pc.d_b); if (pc.d_wu) (void)hipFree(pc.d_wu); if (pc.d_wd) (void)hipFree(pc.d_wd); if (pc.d_wg16) (void)hipFree(pc.d_wg16);
    if (pc.d_wu4) (void)hipFree(pc.d_wu4);
    if (pc.d_wp) (void)hipFree(pc.d_wp);
    delete pc.w1a; pc.w1a = nullptr;
    if (pc.dw.d_w) (void)hipFree(pc.dw.d_w); if (pc.dw.d_b) (void)hipFree(pc.dw.d_b);
  }
  for (auto& kv : c->deconvs) {
    if (kv.second.d_w) (void)hipFree(kv.second.d_w); if (kv.second.d_b) (void)hipFree(kv.second.d_b);
    if (kv.second.d_wd) (void)hipFree(kv.second.d_wd); if (kv.second.d_bz) (void)hipFree(kv.second.d_bz);
    if (kv.second.d_wp) (void)hipFree(kv.second.d_wp);
  }
  c->convs.clear(); c->deconvs.clear();
}

void fisr_pwc_destroy(fisr_pwc* c) {
  if (!c) return;
  DeviceGuard guard(c->dev);
  pwc_release_packed(c);
  delete c;
}

const char* fisr_pwc_last_error(const fisr_pwc* c) { return c ? c->err.c_str() : g_err.c_str(); }

int fisr_pwc_num_variables(void) { return (int)pwc_variable_list().size(); }

// name / shape of variable i (for hosts that enumerate what a checkpoint must hold); returns rank or < 0
int fisr_pwc_variable(int i, const char** name, int64_t* shape4) {
  static const auto list = pwc_variable_list();
  if (i < 0 || i >= (int)list.size() || !name || !shape4) return FISR_EINVAL;
  *name = list[i].first.c_str();
  for (size_t k = 0; k < list[i].second.size(); ++k) shape4[k] = list[i].second[k];
  return (int)list[i].second.size();
}

int fisr_pwc_set_weight(fisr_pwc* c, const char* name, const float* host, const int64_t* shape, int rank) {
  if (!c || !name || !host || !shape) return pfail(c, FISR_EINVAL, "fisr_pwc_set_weight: null argument");
  std::string s(name);
  const size_t colon = s.rfind(':');
  if (colon != std::string::npos) s = s.substr(0, colon);
  auto it = c->vars.find(s);
  if (it == c->vars.end()) return 1;                     // not ours (optimizer slots, global step, ...)
  PwcVar& v = it->second;
  size_t n = 1;
  if (rank != (int)v.shape.size()) return pfail(c, FISR_EINVAL, s + ": wrong rank");
  for (int k = 0; k < rank; ++k) { if (shape[k] != v.shape[k]) return pfail(c, FISR_EINVAL, s + ": wrong shape"); n *= (size_t)shape[k]; }
  v.v.assign(host, host + n);
  v.have = true;
  c->finalized = false;
  return 0;
}

// precision: FISR_PREC_F32W (fp32 tensors and arithmetic; the dense layers on the Winograd kernel) or FISR_PREC_F16 (fp16 feature
// tensors, fp32 accumulation, fp32 flows; the dense layers on the LDS-DMA kernel)
int fisr_pwc_finalize_precision(fisr_pwc* c, int precision) {
  if (!c) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_finalize: ctx is NULL");
  if (precision != FISR_PREC_F32W && precision != FISR_PREC_F32W4 && precision != FISR_PREC_F32 && precision != FISR_PREC_F16)
    return pfail(c, FISR_EINVAL, "fisr_pwc_finalize_precision: FISR_PREC_F32W, FISR_PREC_F32W4 or FISR_PREC_F16");
  for (auto& kv : c->vars) if (!kv.second.have) return pfail(c, FISR_EMISSING, "missing variable " + kv.first);
  DeviceGuard guard(c->dev);
  HIP_OK(nullptr, guard.err);
  pwc_release_packed(c);
  c->precision = precision == FISR_PREC_F16 ? FISR_PREC_F16 : FISR_PREC_F32W;
  c->f4 = precision == FISR_PREC_F32W4;
  int rc = 0;
  const int real[7] = {3, 16, 32, 64, 96, 128, 196};
  for (int l = 1; l <= PWC_LVLS && !rc; ++l) {
    const std::string p = "pwcnet/featpyr/conv" + std::to_string(l);
    rc = pwc_pack_conv(c, p + "a", iota_map(real[l - 1]), PWC_CH[l - 1], c->convs[p + "a"]);
    if (!rc) rc = pwc_pack_conv(c, p + "aa", iota_map(real[l]), PWC_CH[l], c->convs[p + "aa"], true);
    if (!rc) rc = pwc_pack_conv(c, p + "b", iota_map(real[l]), PWC_CH[l], c->convs[p + "b"], true);
  }
  for (int l = PWC_LVLS; l >= PWC_PRED && !rc; --l) {
    const DecLayout L(l);
    const std::string ls = std::to_string(l);
    for (int i = 0; i < 5 && !rc; ++i) {
      const int first = i == 0 ? L.off_corr : L.off_act[i - 1];
      std::vector<int> m = L.map_from(i == 0 ? 5 : i - 1);
      for (int& x : m) x -= first;
      const std::string n = "pwcnet/predict_flow/conv" + ls + "_" + std::to_string(i);
      rc = pwc_pack_conv(c, n, m, L.total - first, c->convs[n], true);
    }
    if (!rc) rc = pwc_pack_conv(c, "pwcnet/predict_flow/flow" + ls, L.map_from(4), L.total, c->convs["pwcnet/predict_flow/flow" + ls], true);
    int ci = 0;
    for (int i = 0; i < 7 && !rc; ++i) {
      const std::string n = "pwcnet/ctxt/dc_conv" + ls + std::to_string(i + 1);
      rc = i == 0 ? pwc_pack_conv(c, n, L.map_from(4), L.total, c->convs[n], true)
                  : pwc_pack_conv(c, n, iota_map(ci), ci, c->convs[n], true);
      ci = PWC_CTXT[i][0];
    }
    if (l != PWC_PRED && !rc) {
      rc = pwc_pack_deconv(c, "pwcnet/upsample/up_flow" + ls, iota_map(2), 4, c->deconvs["pwcnet/upsample/up_flow" + ls]);
      if (!rc) rc = pwc_pack_deconv(c, "pwcnet/upsample/up_feat" + ls, L.map_from(4), L.total, c->deconvs["pwcnet/upsample/up_feat" + ls]);
    }
  }
  if (rc) return rc;
  c->finalized = true;
  return 0;
}
int fisr_pwc_finalize(fisr_pwc* c) { return fisr_pwc_finalize_precision(c, FISR_PREC_F32W); }

// The network alone on a prepared pair: im [2, H, W, 4] device float32 (RGB/255 + zero channel; H, W multiples of 64).
// flow_pred [2, H, W, 2] (direction 0: a->b, 1: b->a) = x4 bilinear * 4 of the level-2 flow (nullable);
// pyr[10] (nullable, entries nullable): refined flows of levels 6..2 for direction 0 then direction 1, [h_l, w_l, 2].
size_t fisr_pwc_nn_workspace_bytes(const fisr_pwc* c, int H, int W) {
  if (!c || !c->finalized || H < 64 || W < 64 || H % 64 || W % 64) return 0;
  return pwc_ws(const_cast<fisr_pwc*>(c), 2, H, W, 2, false);
}

int fisr_pwc_nn(fisr_pwc* c, const float* im, int H, int W, float* flow_pred, float* const* pyr, void* ws, size_t ws_bytes, void* stream) {
  if (!c || !c->finalized) return pfail(c, FISR_ESTATE, "fisr_pwc_nn: weights not finalized");
  if (!im || !ws || H < 64 || W < 64 || H % 64 || W % 64) return pfail(c, FISR_EINVAL, "fisr_pwc_nn: H and W must be multiples of 64");
  if (ws_bytes < fisr_pwc_nn_workspace_bytes(c, H, W)) return pfail(c, FISR_ENOMEM, "fisr_pwc_nn: workspace too small");
  DeviceGuard guard(c->dev);
  HIP_OK(nullptr, guard.err);
  hipStream_t st = (hipStream_t)stream;
  const std::vector<std::pair<int, int>> pairs = {{0, 1}, {1, 0}};
  return with_pwc_elem(c, [&](auto tag) {
    typedef decltype(tag) TE;
    return pwc_run<TE>(c, im, nullptr, 2, H / 2, W / 2, H, W, pairs, pyr, ws, ws_bytes, st, false, nullptr, [&](int k, float* f2) {
      if (!flow_pred) return 0;
      hipLaunchKernelGGL(pwc_upsample4_kernel, dim3(grid_for((size_t)H * W)), dim3(256), 0, st, f2, 4, 0, H / 4, W / 4, flow_pred + (size_t)k * H * W * 2);
      return hipGetLastError() == hipSuccess ? 0 : pfail(c, FISR_EHIP, "fisr_pwc_nn: flow_pred");
    });
  });
}

// What the reference script's loop (:104-141) computes for a run of nframes YUV uint8 frames [h, w, 3] on the device, in ONE call:
// flows [nframes - 1, 2, h, w, 2] float32 LR pixels (pair fr: [0] = fr -> fr+1, [1] = fr+1 -> fr).  Every frame is pre-processed and
// its feature pyramid extracted once (the script's pair-by-pair loop does both twice for the inner frames); the 2 (nframes - 1)
// directions run through the decoder as batches of up to 8.
size_t fisr_pwc_flow_stack_workspace_bytes(const fisr_pwc* c, int nframes, int h, int w) {
  if (!c || !c->finalized || h < 8 || w < 8 || nframes < 2) return 0;
  return pwc_ws(const_cast<fisr_pwc*>(c), nframes, round_up(2 * h, 64), round_up(2 * w, 64), 2 * (nframes - 1), true);
}

int fisr_pwc_flow_stack(fisr_pwc* c, const uint8_t* const* yuv, int nframes, int h, int w, float* flows, void* ws, size_t ws_bytes, void* stream) {
  if (!c || !c->finalized) return pfail(c, FISR_ESTATE, "fisr_pwc_flow_stack: weights not finalized");
  if (!yuv || !flows || !ws || h < 8 || w < 8 || nframes < 2) return pfail(c, FISR_EINVAL, "fisr_pwc_flow_stack: bad argument");
  for (int k = 0; k < nframes; ++k) if (!yuv[k]) return pfail(c, FISR_EINVAL, "fisr_pwc_flow_stack: null frame");
  if (ws_bytes < fisr_pwc_flow_stack_workspace_bytes(c, nframes, h, w)) return pfail(c, FISR_ENOMEM, "fisr_pwc_flow_stack: workspace too small");
  DeviceGuard guard(c->dev);
  HIP_OK(nullptr, guard.err);
  hipStream_t st = (hipStream_t)stream;
  const int H = round_up(2 * h, 64), W = round_up(2 * w, 64);
  std::vector<std::pair<int, int>> pairs;
  for (int fr = 0; fr + 1 < nframes; ++fr) { pairs.push_back({fr, fr + 1}); pairs.push_back({fr + 1, fr}); }
  return with_pwc_elem(c, [&](auto tag) {
    typedef decltype(tag) TE;
    return pwc_run<TE>(c, nullptr, yuv, nframes, h, w, H, W, pairs, nullptr, ws, ws_bytes, st, false, nullptr, [&](int k, float* f2) {
      hipLaunchKernelGGL(pwc_flow_out_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, st, f2, 4, 0, H / 4, W / 4, flows + (size_t)k * h * w * 2, h, w);
      return hipGetLastError() == hipSuccess ? 0 : pfail(c, FISR_EHIP, "fisr_pwc_flow_stack: flow_out");
    });
  });
}

// One iteration of the reference script's loop (:118-140): two YUV uint8 frames [h, w, 3] on the device ->
// flows a->b and b->a in LR pixels, [h, w, 2] float32 each (what the script writes into pred[fr, 0] and pred[fr, 1]).
size_t fisr_pwc_flow_workspace_bytes(const fisr_pwc* c, int h, int w) { return fisr_pwc_flow_stack_workspace_bytes(c, 2, h, w); }

int fisr_pwc_flow_pair(fisr_pwc* c, const uint8_t* yuv_a, const uint8_t* yuv_b, int h, int w, float* flow_ab, float* flow_ba,
                       void* ws, size_t ws_bytes, void* stream) {
  if (!c || !c->finalized) return pfail(c, FISR_ESTATE, "fisr_pwc_flow_pair: weights not finalized");
  if (!yuv_a || !yuv_b || !flow_ab || !flow_ba || !ws || h < 8 || w < 8) return pfail(c, FISR_EINVAL, "fisr_pwc_flow_pair: bad argument");
  if (ws_bytes < fisr_pwc_flow_workspace_bytes(c, h, w)) return pfail(c, FISR_ENOMEM, "fisr_pwc_flow_pair: workspace too small");
  DeviceGuard guard(c->dev);
  HIP_OK(nullptr, guard.err);
  hipStream_t st = (hipStream_t)stream;
  const int H = round_up(2 * h, 64), W = round_up(2 * w, 64);
  const uint8_t* yuv[2] = {yuv_a, yuv_b};
  float* outs[2] = {flow_ab, flow_ba};
  const std::vector<std::pair<int, int>> pairs = {{0, 1}, {1, 0}};
  return with_pwc_elem(c, [&](auto tag) {
    typedef decltype(tag) TE;
    return pwc_run<TE>(c, nullptr, yuv, 2, h, w, H, W, pairs, nullptr, ws, ws_bytes, st, false, nullptr, [&](int k, float* f2) {
      hipLaunchKernelGGL(pwc_flow_out_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, st, f2, 4, 0, H / 4, W / 4, outs[k], h, w);
      return hipGetLastError() == hipSuccess ? 0 : pfail(c, FISR_EHIP, "fisr_pwc_flow_pair: flow_out");
    });
  });
}

// ---- op-level entries (parity tests of the flow network's kernels at the sizes the bench runs them at) ----
// precision FISR_PREC_F32W: float32 tensors; FISR_PREC_F16: fp16 feature tensors (in, and out unless out_f32), float32 `add`.
// One tf.layers.conv2d 3x3 'same' (+ bias, leaky relu, optional add) exactly as the network launches it: the input is the
// channel range [in_co, in_co + cin_buf) of a buffer with pixel stride in_cs, the output the range [out_co, out_co + cout)
// of a buffer with pixel stride out_cs.  w_host: TF HWIO [3,3,ci,cout]; chmap (nullable = identity, then ci == cin_buf):
// buffer channel, relative to in_co, of TF input channel j (the dense blocks' padded channel groups).  route 0: the
// network's own choice, 1: generic implicit GEMM, 2: fp32 Winograd F(2x2), 3: FISRnet's direct kernel, 4: fp16 LDS-DMA kernel, 5: fp32
// Winograd F(4x4) (FISR_PREC_F32W4 only), 6: fp32 pointwise map + 9-tap gather (two output channels, no activation, cin_buf % 32 == 0)
// 7: conv1a (3 -> 16 channels in a 4-wide buffer, stride 2, even sizes)
// (2 - 7: error if the layer is not eligible).  Returns the route taken (1 .. 7) or a negative error.
int fisr_pwc_op_conv(const void* in, int in_cs, int in_co, int cin_buf, const float* w_host, const float* b_host, int ci, int cout,
                     const int* chmap, void* out, int out_f32, int out_cs, int out_co, const float* add, int add_cs, int add_co, int n, int h, int w,
                     int stride, int dil, float slope, int route, int precision, void* stream) {
  if (!in || !w_host || !b_host || !out || ci < 1 || cout < 1 || cin_buf < ci || n < 1 || h < 1 || w < 1 || stride < 1 || stride > 2 || dil < 1)
    return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_conv: bad argument");
  if (precision != FISR_PREC_F32W && precision != FISR_PREC_F32W4 && precision != FISR_PREC_F16)
    return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_conv: FISR_PREC_F32W, FISR_PREC_F32W4 or FISR_PREC_F16");
  if ((in_cs | in_co | out_cs | out_co | cin_buf) & 3) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_conv: channel strides / offsets / cin_buf must be multiples of 4");
  fisr_pwc tmp;
  tmp.dev = device_of(out);
  tmp.precision = precision == FISR_PREC_F32W4 ? FISR_PREC_F32W : precision;
  tmp.f4 = precision == FISR_PREC_F32W4;
  DeviceGuard guard(tmp.dev);
  HIP_OK(nullptr, guard.err);
  PwcVar& kw = tmp.vars["op/kernel"]; PwcVar& kb = tmp.vars["op/bias"];
  kw.shape = {3, 3, ci, cout}; kw.v.assign(w_host, w_host + (size_t)9 * ci * cout);
  kb.shape = {cout}; kb.v.assign(b_host, b_host + cout);
  std::vector<int> m = chmap ? std::vector<int>(chmap, chmap + ci) : iota_map(ci);
  for (int c : m) if (c < 0 || c >= cin_buf) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_conv: chmap entry out of range");
  int rc = pwc_pack_conv(&tmp, "op", m, cin_buf, tmp.convs["op"], route != 1);
  if (rc) { pwc_release_packed(&tmp); return rc; }
  if (route >= 1 && route != 6 && tmp.convs["op"].d_wp) {      // (a forced other kernel: the pointwise route would win the choice)
    (void)hipFree(tmp.convs["op"].d_wp); tmp.convs["op"].d_wp = nullptr;
  }
  if (route >= 1 && route != 7 && tmp.convs["op"].w1a) { delete tmp.convs["op"].w1a; tmp.convs["op"].w1a = nullptr; }
  int took = 0;
  rc = with_pwc_elem(&tmp, [&](auto tag) {
    typedef decltype(tag) TE;
    PwcRunner<TE> r; r.ctx = &tmp; r.st = (hipStream_t)stream;
    took = r.conv_route("op", n, h, w, in_cs, out_cs, out_f32 != 0 && std::is_same<TE, _Float16>::value, stride, dil, slope, add != nullptr);
    if (route >= 2 && took != route) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_conv: the requested kernel does not take this layer");
    void* scratch = nullptr;
    if (took == 6) {     // (the pointwise route stages its 20 tap sums per pixel in the runner's arena)
      const size_t sbytes = (size_t)n * h * w * 20 * sizeof(float) + 512;
      if (hipMalloc(&scratch, sbytes) != hipSuccess) return pfail(nullptr, FISR_EHIP, "fisr_pwc_op_conv: hipMalloc");
      r.ar.base = (char*)scratch; r.ar.cap = sbytes;
    }
    r.conv("op", (const TE*)in, in_cs, in_co, out, out_f32 != 0, out_cs, out_co, n, h, w, stride, dil, slope, add, add_cs, add_co);
    hipError_t e = hipStreamSynchronize(r.st);
    if (scratch) (void)hipFree(scratch);
    if (r.rc) return r.rc;
    return e != hipSuccess ? pfail(nullptr, FISR_EHIP, std::string("fisr_pwc_op_conv: ") + hipGetErrorString(e)) : 0;
  });
  pwc_release_packed(&tmp);
  return rc ? rc : took;
}

// tf.layers.conv2d_transpose(x, 2, 4, 2, 'same') (model_pwcnet.py:1196): in = range [in_co, in_co + cin4) of a buffer with pixel
// stride in_cs, [n,h,w] (in_f32: float32 -- the flow -- else the engine's element type); w_host TF layout [4,4,2,ci]; chmap as
// above; out channels [out_co, out_co + 2) of [n,2h,2w] x out_cs (element type).
int fisr_pwc_op_deconv(const void* in, int in_f32, int in_cs, int in_co, int cin4, const float* w_host, const float* b_host, int ci, const int* chmap,
                       void* out, int out_cs, int out_co, int n, int h, int w, int precision, void* stream) {
  if (!in || !w_host || !b_host || !out || ci < 1 || cin4 < ci || (cin4 & 3) || (in_cs & 3) || (in_co & 3) || n < 1 || h < 1 || w < 1)
    return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_deconv: bad argument");
  if (precision != FISR_PREC_F32W && precision != FISR_PREC_F16) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_deconv: FISR_PREC_F32W or FISR_PREC_F16");
  fisr_pwc tmp;
  tmp.dev = device_of(out);
  tmp.precision = precision;
  DeviceGuard guard(tmp.dev);
  HIP_OK(nullptr, guard.err);
  PwcVar& kw = tmp.vars["op/kernel"]; PwcVar& kb = tmp.vars["op/bias"];
  kw.shape = {4, 4, 2, ci}; kw.v.assign(w_host, w_host + (size_t)32 * ci);
  kb.shape = {2}; kb.v.assign(b_host, b_host + 2);
  std::vector<int> m = chmap ? std::vector<int>(chmap, chmap + ci) : iota_map(ci);
  for (int c : m) if (c < 0 || c >= cin4) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_deconv: chmap entry out of range");
  int rc = pwc_pack_deconv(&tmp, "op", m, cin4, tmp.deconvs["op"]);
  if (!rc)
    rc = with_pwc_elem(&tmp, [&](auto tag) {
      typedef decltype(tag) TE;
      PwcRunner<TE> r; r.ctx = &tmp; r.st = (hipStream_t)stream;
      // (the wide-input path of the fp16 engine stages its 32 tap sums per pixel in the runner's arena)
      void* scratch = nullptr;
      const size_t sbytes = (size_t)n * h * w * 32 * sizeof(TE) + 512;
      if (hipMalloc(&scratch, sbytes) != hipSuccess) return pfail(nullptr, FISR_EHIP, "fisr_pwc_op_deconv: hipMalloc");
      r.ar.base = (char*)scratch; r.ar.cap = sbytes;
      if (in_f32) r.template deconv<float>("op", (const float*)in, in_cs, in_co, (TE*)out, out_cs, out_co, n, h, w);
      else r.template deconv<TE>("op", (const TE*)in, in_cs, in_co, (TE*)out, out_cs, out_co, n, h, w);
      hipError_t e = hipStreamSynchronize(r.st);
      (void)hipFree(scratch);
      return r.rc ? r.rc : (e != hipSuccess ? pfail(nullptr, FISR_EHIP, std::string("fisr_pwc_op_deconv: ") + hipGetErrorString(e)) : 0);
    });
  pwc_release_packed(&tmp);
  return rc;
}

// core_costvol.cost_volume + leaky relu (model_pwcnet.py:1277): c1, c2 dense [n,h,w,c] (c % 4 == 0) -> 81 channels at out_co of a
// buffer with pixel stride out_cs
int fisr_pwc_op_costvol(const void* c1, const void* c2, int c, void* out, int out_cs, int out_co, int n, int h, int w, int precision, void* stream) {
  if (!c1 || !c2 || !out || c < 4 || (c & 3) || (out_cs & 3) || (out_co & 3) || n < 1 || n > PWC_MAX_ITEMS || h < 1 || w < 1)
    return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_costvol: bad argument");
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  const PwcItems id = identity_items(n);
  hipError_t e = precision == FISR_PREC_F16
      ? launch_costvol<_Float16>((const _Float16*)c1, c, 0, id, (const _Float16*)c2, id, c, (_Float16*)out, out_cs, out_co, n, h, w, (hipStream_t)stream)
      : launch_costvol<float>((const float*)c1, c, 0, id, (const float*)c2, id, c, (float*)out, out_cs, out_co, n, h, w, (hipStream_t)stream);
  if (e != hipSuccess) return pfail(nullptr, FISR_EHIP, std::string("fisr_pwc_op_costvol: ") + hipGetErrorString(e));
  return 0;
}

// core_warp.dense_image_warp (model_pwcnet.py:1178): img dense [n,h,w,c] sampled at (x + scale*u, y + scale*v), (u, v) = channels
// f_co, f_co + 1 of a buffer with pixel stride f_cs -> out dense [n,h,w,c]
int fisr_pwc_op_warp(const void* img, int c, const void* flow, int f_cs, int f_co, float scale, void* out, int n, int h, int w, int precision, void* stream) {
  if (!img || !flow || !out || c < 4 || (c & 3) || n < 1 || n > PWC_MAX_ITEMS || h < 2 || w < 2) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_op_warp: bad argument");
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  const PwcItems id = identity_items(n);
  const dim3 grid(grid_for((size_t)n * h * w * c / 4));
  if (precision == FISR_PREC_F16)
    hipLaunchKernelGGL(pwc_warp_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)img, id, c, (const _Float16*)flow, f_cs, f_co, scale, (_Float16*)out, n, h, w);
  else
    hipLaunchKernelGGL(pwc_warp_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)img, id, c, (const float*)flow, f_cs, f_co, scale, (float*)out, n, h, w);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

// test hooks for the two pre/post-processing kernels
int fisr_pwc_prep(const uint8_t* yuv, int h, int w, float* out, int PH, int PW, void* stream) {
  if (!yuv || !out || PH < 2 * h || PW < 2 * w) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_prep: bad argument");
  static const ColorConsts cc = make_color_consts();
  DeviceGuard guard(device_of(out));
  hipLaunchKernelGGL(pwc_prep_kernel<float>, dim3(grid_for((size_t)PH * PW)), dim3(256), 0, (hipStream_t)stream, yuv, h, w, out, PH, PW, cc);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}
int fisr_pwc_flow_out(const float* flow2, int FH, int FW, float* out, int h, int w, void* stream) {
  if (!flow2 || !out || 4 * FH < 2 * h || 4 * FW < 2 * w) return pfail(nullptr, FISR_EINVAL, "fisr_pwc_flow_out: bad argument");
  DeviceGuard guard(device_of(out));
  hipLaunchKernelGGL(pwc_flow_out_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, (hipStream_t)stream, flow2, 2, 0, FH, FW, out, h, w);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

}  // extern "C"
