// 3x3 SAME convolution in fp32 by Winograd minimal filtering F(4x4, 3x3) on the gfx950 matrix cores (round 3).
//
// Same operator and fused neighbours as conv3x3_wino8p.h (reference ops.py:7-11 + relu-on-load / relu / residual / dual-source
// concat / depth_to_space), same NHWC fp32 tensors -- the THIRD algorithm of the fp32 engine:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 4x4 output tile, 6x6 input patch d, 3x3 filter g
//
// 36 multiplies per 16 outputs = 2.25 per output instead of F(2x2)'s 4 and the direct algorithm's 9: the fp32 MFMA pipe does a
// QUARTER of the direct work (the kernel needs the pipe busy 0.45 of the time to match conv3x3_wino8p.h at 0.70).  Everything is
// fp32: transforms are fp32 fma chains, products accumulate on v_mfma_f32_16x16x4_f32, U = G g G^T is computed on the host in
// double and rounded once.  The price is the conditioning of the F(4,3) transforms (interpolation points 0, +-1, +-2, inf): about
// ten times F(2x2)'s rounding error per convolution (2e-5 instead of 2e-6 on O(1) outputs).  Through the network it does not add
// up -- 1.5e-6 against the fp64 oracle on the full tile, 1.1-1.6e-6 on the undamped and the harsh weight sets, the F(2x2)
// engine's figures -- but it is its own engine id (FISR_PREC_F32W4) all the same, FISR_PREC_F32W stays what it was.
//
// GEMM view, per transform position p = 0..35:  M_p[co][tile] = sum_ci U_p[co][ci] * V_p[ci][tile].
// Work item = 16 x 32 output pixels (4 x 8 tiles of 4 x 4) x 64 output channels; 512 threads = 8 waves = two per SIMD.
//   * A wave owns ALL 36 positions of a 16-channel x 16-tile block (36 accumulators of v_mfma_f32_16x16x4_f32 = 144 registers):
//     the output transform A^T M A then happens in the wave's own registers -- no exchange between waves, no LDS round trip, no
//     transposition (conv3x3_wino8p.h splits the positions over two waves and spends 10-14 % of a 64-channel item on that
//     exchange and on the DPP transposes behind it).  The accumulator layout of the 16x16 MFMA gives a lane 4 consecutive channels
//     of one tile; one lane transposition (ds_bpermute_b32) puts the four lanes of a 64-byte record next to each other.
//   * K loop over 4-channel chunks (= the K of one MFMA), ONE barrier per chunk; LDS (151 552 B):
//       U[2]    36 positions x 64 channels x 4 ci       LDS-DMA of the host-made slab (the LDS image), one chunk ahead
//       V[2]    36 positions x 32 tiles x 4 ci          B^T d B of the NEXT chunk, computed by waves 0-3 while everybody multiplies
//       RAW[2]  18 x 34 halo pixels x 32 B              LDS-DMA from the activation tensor: PAIRS of chunks, one pair ahead
//     Every global byte goes global -> LDS by `buffer_load_dwordx4 ... lds` from inline asm (see conv3x3_dma.h: hidden from the
//     compiler, counted by hand; a lane whose offset lies behind the buffer's end writes ZEROS -- the zero padding is free).
//   * Fragments are 16-byte reads of four consecutive positions (9 + 9 ds_read_b128 feed the 36 MFMAs of a chunk): U as
//     [position quad][channel quarter][k][16 channels][4], V as [position quad][tile half][slot(k, tile)][4] with
//     slot = 16 k + (tile ^ 2 k): the transform lanes (2 tiles x 4 channels per ds_write_b128 phase) and the MFMA lanes (the
//     lane groups of a ds_read_b128 phase: 8 tiles of one k and the 8 other tiles of the next) both hit every bank once.
//   * A raw record is 32 bytes = a PAIR of chunks of one pixel, fetched by two neighbouring lanes as 32 contiguous bytes (a copy
//     instruction then touches 32 memory lines, not 64: that count is what a raw copy costs).  The halo rows hold their columns
//     grouped by column mod 4 and the two halves of a record trade places with bit 2 of the halo row, so a ds_read_b32 phase of the
//     transform -- 4 tiles of one tile row, the 4 below them, 4 channels -- reads 32 different banks.
//   * The fp32 MFMA and the vector ALU of a SIMD do not overlap (probed), so every vector instruction is paid on top of the MFMAs:
//     the transforms are packed fp32 (op_sel: 6 instructions per 6 -> 6 row), the bias rides in accumulator (1,1), the relu of
//     relu-on-load layers is applied once per element in LDS, record offsets ride in the scalar operand of the buffer instructions.
//   * Persistent (one workgroup per CU walks its work items, copy and transform streams run on into the next item); the 2x2 max
//     pooling behind an encoder level (ops.py:54) is a second store of that level's last convolution (POOL instantiation).
#pragma once
#include <algorithm>
#include <type_traits>
#include <vector>
#include "conv3x3.h"

namespace fisr {

typedef __attribute__((address_space(3))) void* wf4_lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int F4_TH = 16, F4_TW = 32;                  // output pixels of a work item
constexpr int F4_HH = F4_TH + 2, F4_HW = F4_TW + 2;    // halo tile 18 x 34
constexpr int F4_HALO = F4_HH * F4_HW;                 // 612 pixels (16 bytes each per chunk, 32 per raw pair)
constexpr int F4_CH = 4;                               // channels per K chunk
constexpr int F4_BN = 64;                              // output channels per work item
constexpr int F4_RAW_COPIES = 10;                      // 1 KB LDS-DMA copies per raw chunk (640 slots, 612 used); a raw PAIR is two of these
constexpr int F4_RAW_BYTES = F4_RAW_COPIES * 1024;     // 10240
constexpr int F4_U_BYTES = 36 * F4_BN * F4_CH * 4;     // 36864: one weight slab = 36 copies
constexpr int F4_V_BYTES = 36 * 32 * F4_CH * 4;        // 18432
// Position (i, j) of the 6 x 6 transform grid lives in slot 6 i + F4_PERM[j] of V, U and the accumulators: the order in which the
// packed horizontal pass of the input transform leaves its outputs (t0 t5 t1 t3 t2 t4).
__host__ __device__ constexpr int wf4_slot(int i, int j) { return 6 * i + (j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 4 : j == 3 ? 3 : j == 4 ? 5 : 1); }
constexpr size_t wf4_lds_bytes() { return (size_t)2 * F4_U_BYTES + 2 * F4_V_BYTES + 2 * (2 * F4_RAW_BYTES); }   // 151552
// UPS (fused x2 bilinear): two staging buffers for the HALF-resolution halo of a raw pair -- 10 x 18 pixels (192 slots) in two
// 16-byte planes, one per chunk of the pair -- behind the raw buffers: 163840 bytes, all of a CU's LDS
constexpr int F4_L_ROWS = 10, F4_L_COLS = 18, F4_L_PLANE = 3072, F4_L_BYTES = 2 * F4_L_PLANE;
constexpr size_t wf4_lds_bytes_ups() { return wf4_lds_bytes() + 2 * F4_L_BYTES; }

// FISR_F4ABL: performance-diagnosis ablations (WRONG results; scripts/probes/wf4_bench.hip): 1 no weight copies in the K loop,
// 8192 SHARE: no V stores, 16384 SHARE: no V copies, 32768 SHARE: the V scratch of a workgroup is four chunks (what an L2-resident V would cost),
// 4 no input transform, 16 no MFMAs, 128 workgroups de-phased at start, 256 wait for the stores behind the epilogue, 1024 no barrier, 2048 no vmcnt wait at the end of an iteration, 512 one store
// per lane instead of 16 (earlier versions: 2 no raw copies, 32 / 64 raw access patterns)
#ifndef FISR_F4ABL
#define FISR_F4ABL 0
#endif
// per-workgroup / per-wave timestamps (p.trace): compiled in for scripts/probes/wf4_bench.hip only (-DFISR_F4_TRACE=1)
#ifndef FISR_F4_TRACE
#define FISR_F4_TRACE 0
#endif
#ifndef FISR_F4_STORE_AUX
#define FISR_F4_STORE_AUX 0      // cache policy of the output stores (A/B hook; 2 = nontemporal: 1-2 % slower on the residual layers)
#endif
#define FISR_F4_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
// schedule hooks (A/B): quad behind which the transform's vertical pass / first row runs, first quad of the weight copies
#ifndef FISR_F4_COLQ
#define FISR_F4_COLQ 1
#endif
#ifndef FISR_F4_ROWQ
#define FISR_F4_ROWQ 2
#endif
#ifndef FISR_F4_UQ
#define FISR_F4_UQ 0
#endif
// the plain residual instantiation issues ALL 36 weight copies of a chunk from the copy waves (A/B hook: 0 = the 4 + 5 split of the others)
#ifndef FISR_F4_SOFF_RUN
#define FISR_F4_SOFF_RUN 1       // the stores' uniform offsets as running scalar sums (A/B hook: 0 = sixteen hoisted products)
#endif
// FISR_F4_RAWEARLY (r04): the raw pair is requested at the START of its odd iteration instead of behind that iteration's weight
// copies (+0.6 of an iteration of latency budget: 1.5-2.0 instead of 1.0-1.4 iterations until it is needed).  vmcnt is one
// in-order counter, so the weight copies of an odd iteration then all come from the transform waves (9 each) and those of an even
// one from the copy waves (9 each): a wave that waits for weight copies never has an older raw copy in front of them.
#ifndef FISR_F4_RAWEARLY
#define FISR_F4_RAWEARLY 0
#endif
#ifndef FISR_F4_PADSKIP
#define FISR_F4_PADSKIP 1      // GENERAL: padding waves of a <= 32-channel layer skip their MFMAs (A/B hook: 0)
#endif
#ifndef FISR_F4_UALL
#define FISR_F4_UALL 1
#endif
#ifndef FISR_F4_U_AUX
#define FISR_F4_U_AUX ""         // cache-policy bits of the weight copies (A/B hook: " nt", " sc1", " sc0 sc1")
#endif
// SHARE: cache policy of the V stores (aux bits of the builtin: 1 sc0, 2 nt, 16 sc1) and of the V copies (A/B hooks)
#ifndef FISR_F4_VST_AUX
#define FISR_F4_VST_AUX 0
#endif
#ifndef FISR_F4_VLD_AUX
#define FISR_F4_VLD_AUX " sc1"
#endif
#ifndef FISR_F4_R_AUX              // ... of the raw copies (A/B hook, -DFISR_F4_R_AUXN=1 nt, 2 sc1, 3 sc0 sc1)
#if FISR_F4_R_AUXN == 1
#define FISR_F4_R_AUX " nt"
#elif FISR_F4_R_AUXN == 2
#define FISR_F4_R_AUX " sc1"
#elif FISR_F4_R_AUXN == 3
#define FISR_F4_R_AUX " sc0 sc1"
#else
#define FISR_F4_R_AUX ""
#endif
#endif
#define FISR_F4_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen" FISR_F4_R_AUX " lds\n\t"
#define FISR_F4_COPYU(OFF, RS, SO) "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen" FISR_F4_U_AUX " lds\n\t"
#define FISR_F4_END(KEEP)          "s_mov_b32 m0, %[" #KEEP "]"

// B^T of F(4,3), in place on six values (rows of [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1])
__device__ __forceinline__ void wf4_bt(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5) {
  const float t0 = fmaf(4.f, x0, fmaf(-5.f, x2, x4));
  const float a = fmaf(-4.f, x2, x4), b = fmaf(-4.f, x1, x3);
  const float c = x4 - x2, e = 2.f * (x3 - x1);
  const float t5 = fmaf(4.f, x1, fmaf(-5.f, x3, x5));
  x0 = t0; x1 = a + b; x2 = a - b; x3 = c + e; x4 = c - e; x5 = t5;
}
// A^T of F(4,3): six values -> four  (rows of [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1])
__device__ __forceinline__ void wf4_at(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y0 = (m0 + s1) + s2;
  y1 = fmaf(2.f, d2, d1);
  y2 = fmaf(4.f, s2, s1);
  y3 = fmaf(8.f, d2, d1) + m5;
}

// 16-byte buffer store (lane offset + uniform offset in the soffset field) with one wait state pinned behind it: see the epilogue.
// (The store stays the compiler's builtin: the soffset registers are SGPR spills reloaded by v_readlane, and "VALU writes SGPR ->
// VMEM reads it" wants 5 wait states that only the hazard recogniser inserts -- as inline asm the store went out one slot behind
// its v_readlane and landed anywhere.  The two scheduling barriers keep everything else out from between store and s_nop.)
typedef unsigned int wf4_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wf4_u32x2 __attribute__((ext_vector_type(2)));
template <int AUX = 0>
__device__ __forceinline__ void wf4_store16(wf4_u32x4 data, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(data, rs, voff, soff, AUX);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 0" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int AUX = 0>
__device__ __forceinline__ void wf4_store8(wf4_u32x2 data, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(data, rs, voff, soff, AUX);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 0" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// POOL: p.pool_out gets the 2x2 max pooling of the output as a second store of the epilogue (its own instantiation: the sixteen
// registers of the pooled pixels cost the other layers 1.5-3 % in spills around the output stage).
// UPS: p.in0 is the HALF-resolution map [N, H/2, W/2, C0] and the convolution runs on its x2 bilinear enlargement (ops.py:69,
// tf.image.resize_images(BILINEAR), the TF-1.13 legacy kernel of glue_kernels.h upsample2) without that tensor ever existing: the
// copy waves fetch the 10 x 18 half-resolution pixels under a halo tile (a quarter of the bytes and a third of the requests of the
// full-resolution halo) into a staging buffer and blend them into the raw pair buffer where the relu-on-load pass of the other
// instantiations sits -- one 2 x 2 output quad per lane, out[2i+1] = x[i] + (x[i+1] - x[i]) / 2, out[2i+2] = x[i+1], the same
// operations in the same order as the stand-alone kernel (fma(d, 0.5, a) rounds like a + d * 0.5: the product is exact).
// GENERAL (r04; PWC-Net's dense and context layers, as conv3x3_wino8p.h's flag of the same name): input and output are channel
// RANGES of wider buffers (pixel strides p.in0_cs / p.rec_cs, first output channel p.rec_co; the pointers carry the input
// offset), Cout is padded to the 64-channel block (channels >= Cout are computed from zero weights and not stored), relu is leaky
// (max(v, slope v)), and the convolution may be DILATED: a dilated 3x3 convolution is an ordinary one inside each of the d x d
// interleaved sub-images (pixel (y, x) belongs to sub-image (y % d, x % d)), zero padding included, so tiles and all tile
// coordinates live in a sub-image and only the addresses are scaled back.  Its own instantiation: one source, no relu-on-load,
// no residual, no pooled store, no fused bilinear, no depth_to_space.
// SHARE (r05): the N blocks of a pixel tile stop redoing the input transform.  p.share consecutive N blocks of a tile form a RUN that
// one workgroup walks item by item; the run's first item (the producer, "P") works as every item of the other instantiations does
// and additionally stores each chunk's V = B^T d B -- the LDS image, 18 KB -- to its workgroup's slice of p.vscr; the other items
// (consumers, "C") fetch V from there by LDS-DMA two iterations ahead and run neither raw copies nor relu pass nor transform.  Same
// V, same MFMAs in the same order: the results are bit-identical to the plain instantiation's (asserted by the tests).
//   * C items keep a ring of FOUR V buffers: chunk c lives in slot (c + 2) & 3, slots 0 / 1 = V[0] / V[1], slots 2 / 3 = the two raw
//     pair buffers (idle in a C item) -- with whole quadruples of chunks per item (launcher: nch % 4 == 0, nch >= 8) the streams of
//     consecutive items meet without a collision in all three transitions (P -> C, C -> C, C -> P; worked through at k_iter).
//   * Who does what is decided per iteration by three uniform flags (tr_on: chunk k + 1 is transformed here; vc_on: chunk k + 2 is
//     copied from p.vscr; rp_on: the raw stream -- requests in odd, relu pass in even iterations -- runs), each looking at the item
//     its target chunk belongs to; all 36 weight copies of a chunk come from the copy waves (the UALL split), the V copies from
//     the transform waves (pieces w + 4 j; 5 / 5 / 4 / 4), so every wave's counted wait sees one kind of copy.
//   * p.vscr is written and read by the SAME workgroup only (write-through to L2, read back by copies that carry sc1, i.e. that
//     do not trust the CU's vector cache), a run apart at least: an s_waitcnt of the storing wave and a barrier lie in between.
// MEASURED AND NOT IN THE PRODUCT (profiles/r05_wf4_vshare_ab.txt): bit-identical on every shape, consumers 24 % faster
// per chunk (2.9 k against 3.8 k cycles), and every layer SLOWER as a whole (64->256 at 544x992: +10..15 %; 128->128, 256->256:
// +15..18 %): 18 KB of V per chunk and N block through the CU's memory pipeline -- 20 store and 18 copy instructions -- cost what
// the transform they replace costs, even when V never leaves L2 (ablation 32768: break-even at best); with no V traffic at all
// (ablations 8192 + 16384, wrong results) the same launches run 9-13 % faster than the product: that is all the redundant
// transforms cost.  The instantiations are compiled into scripts/probes/wf4_bench (-DFISR_F4_SHARE, `WF4_SHARE=n wf4_bench check`)
// and nowhere else.
// In the product build SHARE is not even a template parameter: a constant false, every branch on it compiles out.
#ifdef FISR_F4_SHARE
template <bool RELU_IN, bool HAS_RES, bool POOL = false, bool UPS = false, bool GENERAL = false, bool SHARE = false>
__global__ __launch_bounds__(512) void conv3x3_wf4_kernel(const ConvArgs p, const int n_items) {
#else
template <bool RELU_IN, bool HAS_RES, bool POOL = false, bool UPS = false, bool GENERAL = false>
__global__ __launch_bounds__(512) void conv3x3_wf4_kernel(const ConvArgs p, const int n_items) {
  constexpr bool SHARE = false;
#endif
  static_assert(!GENERAL || (!RELU_IN && !HAS_RES && !POOL && !UPS), "GENERAL is the plain instantiation on channel ranges");
  static_assert(!SHARE || (!UPS && !GENERAL), "V sharing: the plain / relu-on-load / residual / pooling instantiations");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sU = smem;
  char* const sV = smem + 2 * F4_U_BYTES;
  char* const sR = smem + 2 * F4_U_BYTES + 2 * F4_V_BYTES;
  char* const sL = smem + wf4_lds_bytes();         // (UPS only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- PERSISTENT: one workgroup per CU walks the work items blockIdx.x, + gridDim.x, ... and its copy / transform streams
  //      simply run on into the next item (the last iterations of an item already request and transform the first chunks of
  //      the next one): only a workgroup's first item pays the prologue, and no item waits for a workgroup launch.
  //      Work item id -> (pixel tile, N block), XCD-aware as in conv3x3_wino8p.h: workgroup w runs on XCD w % 8, gridDim.x is a
  //      multiple of 8, each XCD walks a contiguous range of virtual ids in which the N blocks of one pixel tile are neighbours.
  const int dil = GENERAL ? p.dil : 1;                  // (a compile-time 1 for FISRnet: no index arithmetic is added)
  const int in0_cs = GENERAL ? p.in0_cs : p.C0;
  const int rec_cs = GENERAL ? p.rec_cs : p.Cout, rec_co = GENERAL ? p.rec_co : 0;
  const int tiles_x = ((p.W + dil - 1) / dil + F4_TW - 1) / F4_TW, tiles_y = ((p.H + dil - 1) / dil + F4_TH - 1) / F4_TH;
  const int nblocks = p.CoutPad / F4_BN;
  // SHARE: the virtual ids are RUNS (share consecutive N blocks of a tile); item_of gives a run's first item
  const int share = SHARE ? p.share : 1;
  const int n_units = SHARE ? n_items / share : n_items;
  const int upt = SHARE ? nblocks / share : nblocks;    // runs (items) per pixel tile
  struct Item { int x0, y0, nb, nblk, ry, rx; };        // (x0, y0): inside sub-image (ry, rx) of image nb
  auto item_of = [&](int b) __attribute__((always_inline)) {
    const int q = n_units >> 3, r = n_units & 7;
    const int xcd = b & 7, loc = b >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int t = v / upt;
    Item it;
    it.nblk = (v - t * upt) * share;
    const int tx = t % tiles_x; t /= tiles_x;
    it.x0 = tx * F4_TW;
    it.y0 = (t % tiles_y) * F4_TH;
    t /= tiles_y;
    it.ry = 0; it.rx = 0;
    if (GENERAL) {
      const int sub = t % (dil * dil);
      t /= dil * dil;
      it.ry = sub / dil; it.rx = sub - it.ry * dil;
    }
    it.nb = t;
    if (FISR_F4ABL & 4096) { it.x0 = (tx & 1) * F4_TW; it.y0 = 0; it.nb = 0; }      // (ablation 4096: every item works on two tiles of image 0 -- no DRAM traffic)
    return it;
  };
  int b_cur = blockIdx.x;
  Item cur = item_of(b_cur);
  int j_cur = 0;                                        // SHARE: position of the current item in its run (0: the producer)
  bool has_next = SHARE ? true : b_cur + (int)gridDim.x < n_items;
  bool nxt_p = !SHARE;                                  // SHARE: the next item opens a run (is a producer)
  Item nxt = SHARE ? cur : has_next ? item_of(b_cur + gridDim.x) : cur;
  if constexpr (SHARE) nxt.nblk = cur.nblk + 1;         // (share >= 2: the launcher's rule)
  const int nch0 = p.C0 / F4_CH, nch = (p.C0 + p.C1) / F4_CH;       // (the launcher guarantees nch >= 4)

  if (FISR_F4ABL & 128) { for (int i = 0; i < (int)((blockIdx.x * 7u) & 31u); ++i) __builtin_amdgcn_s_sleep(4); }     // ablation: de-phase the workgroups
  // Static priority 1 for the copy waves where they carry the most beside their MFMAs -- the bilinear blend (3.3 % faster inside the
  // network, same box, two rounds) and all 36 weight copies of the plain residual instantiation (0.8 %); on the relu-on-load and
  // plain instantiations the same costs 3.7 % / 4.7 %, and priority for the transform waves changes nothing anywhere.
#ifndef FISR_F4_PRIO
  if ((UPS || (FISR_F4_UALL && HAS_RES && !RELU_IN && !POOL && !SHARE)) && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef FISR_F4_PRIO
  if (FISR_F4_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);       // A/B: static priority for the younger / the older half
  if (FISR_F4_PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(1);
  if (FISR_F4_PRIO == 3 && wave >= 4) __builtin_amdgcn_s_setprio(3);
#endif
  unsigned long long t_start = 0, t_first = 0, t_main = 0, t_end1 = 0, t_real = 0;
  if (FISR_F4_TRACE && p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  // =========================== copies (LDS-DMA, 1 KB per wave instruction) ===========================
  // weight copy c (0..35) is linear: wave w issues copies w, w + 8, w + 16, w + 24 and, waves 4-7, copy 28 + w.
  // raw copy c (0..19) of a raw pair moves halo slots 32 c .. 32 c + 31 (two lanes per slot); wave 4 + cw issues copies cw, cw + 4, .., cw + 16.
  // (Measured and dropped: all weight copies on waves 0-3 and the raw copies alone on waves 4-7, so that the wait for the weights
  // -- vmcnt is ONE in-order counter per wave -- does not also wait for the older raw copies: no gain, the transform waves carry
  // the longer instruction stream already.)
  // slot s -> halo row s / 34, column from the position inside the row: columns 0,4,..,32 | 1,5,..,33 | 2,..,30 | 3,..,31.
  const int cw = wave - 4;
  constexpr unsigned OOB = 0x80000000u;
  const size_t img_px = (size_t)p.H * p.W;
  // the raw stream's item: pixel index of this lane's halo slots (or -1: outside the image), the image's buffer resources
  // raw pieces of this wave (4-7): piece cw + 4 j, j = 0..4, of the 20 1-KB pieces of a raw PAIR (two consecutive 4-channel chunks,
  // 32 bytes per pixel).  Lanes 2 i, 2 i + 1 of a piece fetch the two chunks of ONE pixel -- 32 contiguous bytes, 32 instead of 64
  // memory lines per instruction (that count is what a raw copy costs: 0.6k cycles per chunk with one lane per pixel) -- into the
  // two 16-byte halves of the pixel's LDS record, half = chunk ^ f(pixel), f = bit 2 of the halo row: the record halves alternate
  // between the two tile rows of a transform read group, which keeps its 8 tiles x 4 channels on 32 different banks.
  int rpix[5] = {-1, -1, -1, -1, -1};              // pixel index of the lane's halo slot per piece, or -1 (outside the image)
  unsigned rsub = 0;                               // bit j: which chunk of the pair the lane fetches in piece j
  __amdgpu_buffer_rsrc_t rs0, rs1;
  auto raw_geom = [&](const Item& it) __attribute__((always_inline)) {
    rsub = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      int l = lane;
      asm volatile("" : "+v"(l));                  // (recomputed per item: hoisted, the halo coordinates would stay live across the K loops)
      const int s = 32 * (cw + 4 * q) + (l >> 1);
      const int py = s / F4_HW, r = s - py * F4_HW;
      const int px = r < 9 ? 4 * r : r < 18 ? 4 * (r - 9) + 1 : r < 26 ? 4 * (r - 18) + 2 : 4 * (r - 26) + 3;
      const int gy = (it.y0 - 1 + py) * dil + it.ry, gx = (it.x0 - 1 + px) * dil + it.rx;      // (a halo pixel left of / above sub-image row 0 is negative for every ry, rx < dil)
      const bool ok = s < F4_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      rpix[q] = ok ? gy * p.W + gx : -1;
      rsub |= (unsigned)((l & 1) ^ ((py >> 2) & 1)) << q;
    }
    rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.in0 + (size_t)it.nb * img_px * in0_cs), 0, (unsigned)(img_px * in0_cs * 4), 0x00020000);
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? (const float*)p.in1 + (size_t)it.nb * img_px * p.C1 : (const float*)p.in0), 0,
                                            (unsigned)(img_px * (p.in1 ? p.C1 : p.C0) * 4), 0x00020000);
  };
  // UPS: the lane's staging slots instead.  Wave-piece wp = cw (piece 0) and, waves 4-5, 4 + cw (piece 1) of the six 1-KB pieces
  // of a staged pair: plane wp / 3 (= chunk of the pair), slots 64 (wp % 3) + lane; slot -> half-resolution pixel
  // (y0 / 2 - 1 + slot / 18, x0 / 2 - 1 + slot % 18), clamped into the map (the bilinear's edge rule; what lies outside the
  // ENLARGED map is zeroed when the quads are written).
  unsigned lro[2] = {OOB, OOB};
  auto ups_geom = [&](const Item& it) __attribute__((always_inline)) {
    const int hl = p.H >> 1, wl = p.W >> 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int wp = j == 0 ? cw : 4 + cw;
      const int plane = wp >= 3 ? 1 : 0;
      const int s = (wp - 3 * plane) * 64 + l;
      const int ly = s / F4_L_COLS, lx = s - ly * F4_L_COLS;
      const int sy = min(max((it.y0 >> 1) - 1 + ly, 0), hl - 1), sx = min(max((it.x0 >> 1) - 1 + lx, 0), wl - 1);
      lro[j] = s < F4_L_ROWS * F4_L_COLS ? (unsigned)(sy * wl + sx) * (unsigned)(p.C0 * 4) + (unsigned)plane * 16u : OOB;
    }
    // (no size check in this resource -- every offset is clamped into the map, the unused slots carry the out-of-range marker 2^31:
    //  a size derived from p.C0 * 4 is computed on the vector ALU beside the offsets above, which turns the whole resource into
    //  vector registers that the copies' "s" operands cannot take)
    rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.in0 + (size_t)it.nb * ((size_t)hl * wl) * p.C0), 0, 0x7fffffffu, 0x00020000);
  };
  if constexpr (UPS) ups_geom(cur); else raw_geom(cur);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (unsigned)((size_t)nch * nblocks * F4_U_BYTES), 0x00020000);
  const unsigned raw_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sR + (unsigned)cw * 1024u;
  const unsigned u_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sU;
  const unsigned u_voff = (unsigned)lane * 16u;
  // the workgroups of an XCD multiply the same slab at the same time: each walks its 36 pieces from another starting point, so
  // that they do not all ask the same L2 channel for the same lines at once (1-2 % on the 256 / 512-channel layers)
  const unsigned u_rot = ((blockIdx.x >> 3) * 7u) % 36u;

#define FISR_F4_DMA1(RS, VOFF, SOFF, LDS)                                                                       \
  do {                                                                                                          \
    unsigned keep_;                                                                                             \
    asm volatile(FISR_F4_BEGIN(keep, lds) FISR_F4_COPY(o, rs, so) FISR_F4_END(keep)                             \
                 : [keep] "=&s"(keep_) : [rs] "s"(RS), [lds] "s"(LDS), [o] "v"(VOFF), [so] "s"(SOFF) : "memory", "scc"); \
  } while (0)
  // weight copy j (0..3; 4: waves 4-7 only) of chunk kc of N block nblk into U[buf]
  // (UALL: all 36 on waves 4-7, copy cw + 4 j, j = 0..8: the transform waves then have no memory instruction and no vmcnt wait in
  //  the K loop at all, and the instantiation stops spilling.  Only where the copy waves have nothing else to do -- the plain
  //  residual layers: 3.4 % faster inside the network (same box, two rounds: 50.4 -> 48.7 ms per 66 launches).  With the relu pass
  //  on the copy waves it costs 2.4 % (87.8 -> 89.9 ms per 94 launches), with the blend of the fused bilinear 3-4 %, with the
  //  pooling epilogue 1 %, plain without residual nothing; splits that left the transform waves 2-5 copies each: +-1 %.)
  constexpr bool UALL = (FISR_F4_UALL && HAS_RES && !RELU_IN && !POOL && !UPS) || SHARE;
  constexpr bool RAWE = FISR_F4_RAWEARLY && !UALL && !UPS;
  auto copy_u1 = [&](int nblk, int kc, int buf, int j) __attribute__((always_inline)) {
    unsigned c = (UALL ? (unsigned)(cw + 4 * j) : j < 4 ? (unsigned)(wave + 8 * j) : (unsigned)(28 + wave)) + u_rot;
    c = c >= 36u ? c - 36u : c;
    const unsigned so = (unsigned)(((size_t)kc * nblocks + nblk) * F4_U_BYTES) + c * 1024u;
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)F4_U_BYTES + c * 1024u;
    {
      unsigned keep_;
      asm volatile(FISR_F4_BEGIN(keep, lds) FISR_F4_COPYU(o, rs, so) FISR_F4_END(keep)
                   : [keep] "=&s"(keep_) : [rs] "s"(rsw), [lds] "s"(lds), [o] "v"(u_voff), [so] "s"(so) : "memory", "scc");
    }
  };
  // RAWEARLY: weight copy j (0..8) of the nine this wave issues in "its" iterations: piece (wave & 3) + 4 j
  auto copy_u9 = [&](int nblk, int kc, int buf, int j) __attribute__((always_inline)) {
    unsigned c = (unsigned)((wave & 3) + 4 * j) + u_rot;
    c = c >= 36u ? c - 36u : c;
    const unsigned so = (unsigned)(((size_t)kc * nblocks + nblk) * F4_U_BYTES) + c * 1024u;
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)F4_U_BYTES + c * 1024u;
    {
      unsigned keep_;
      asm volatile(FISR_F4_BEGIN(keep, lds) FISR_F4_COPYU(o, rs, so) FISR_F4_END(keep)
                   : [keep] "=&s"(keep_) : [rs] "s"(rsw), [lds] "s"(lds), [o] "v"(u_voff), [so] "s"(so) : "memory", "scc");
    }
  };
  // byte offsets of this lane's raw pixels (its half of the pair included) in the concat source the raw stream is in: recomputed
  // where the source or the item changes, not per copy (every vector instruction of the K loop is paid in full)
  unsigned ro[5] = {OOB, OOB, OOB, OOB, OOB};
  bool ro_first = true;
  auto raw_offsets = [&](bool first) __attribute__((always_inline)) {
    const unsigned csb = (unsigned)(first ? in0_cs : p.C1) * 4u;
#pragma unroll
    for (int j = 0; j < 5; ++j) ro[j] = rpix[j] < 0 ? OOB : (unsigned)rpix[j] * csb + ((rsub >> j) & 1u) * 16u;
    ro_first = first;
  };
  // raw piece j (0..4) of pair pc (chunks 2 pc, 2 pc + 1 of the raw stream's item) into pair buffer pb
  auto copy_pair1 = [&](int pc, int pb, int j) __attribute__((always_inline)) {
    const bool first = 2 * pc < nch0;
    const unsigned so = (unsigned)(first ? pc : pc - (nch0 >> 1)) * 32u;
    const unsigned lds = raw_lds0 + (unsigned)pb * (unsigned)(2 * F4_RAW_BYTES) + (unsigned)j * 4096u;
    const unsigned o = j == 0 ? ro[0] : j == 1 ? ro[1] : j == 2 ? ro[2] : j == 3 ? ro[3] : ro[4];
    if (first) FISR_F4_DMA1(rs0, o, so, lds); else FISR_F4_DMA1(rs1, o, so, lds);
  };
  // SHARE: this workgroup's slice of p.vscr holds the V of one item's whole K, chunk after chunk, each the 18 KB LDS image.
  // V slot s of a consumer's ring: 0 / 1 = V[0] / V[1], 2 / 3 = the raw pair buffers.
  const size_t vscr_bytes = (size_t)nch * F4_V_BYTES;
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(
      SHARE ? (void*)((char*)p.vscr + (size_t)blockIdx.x * vscr_bytes) : (void*)p.wpk, 0, (unsigned)vscr_bytes, 0x00020000);
  auto ring_off = [](int s) { return s < 2 ? s * F4_V_BYTES : 2 * F4_V_BYTES + (s - 2) * (2 * F4_RAW_BYTES); };     // relative to sV
  const unsigned v_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sV;
  // V piece (wave & 3) + 4 j (j = 0..3; j = 4: piece 16 + wave, waves 0-1) of chunk c into ring slot s  (transform waves)
  auto copy_v1 = [&](int c, int slot, int j) __attribute__((always_inline)) {
    const unsigned piece = j < 4 ? (unsigned)(wave + 4 * j) : (unsigned)(16 + wave);
    const unsigned so = (unsigned)((FISR_F4ABL & 32768) ? c & 3 : c) * (unsigned)F4_V_BYTES + piece * 1024u;      // (ablation 32768: four chunks of scratch per workgroup -- V stays in L2)
    const unsigned lds = v_lds0 + (unsigned)ring_off(slot) + piece * 1024u;
    {
      unsigned keep_;
      if (!(FISR_F4ABL & 16384))
      asm volatile(FISR_F4_BEGIN(keep, lds) "buffer_load_dwordx4 %[o], %[rs], %[so] offen" FISR_F4_VLD_AUX " lds\n\t" FISR_F4_END(keep)
                   : [keep] "=&s"(keep_) : [rs] "s"(rsv), [lds] "s"(lds), [o] "v"(u_voff), [so] "s"(so) : "memory", "scc");
    }
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(FISR_F4ABL & 1024)) __builtin_amdgcn_s_barrier();       // (ablation 1024: no barrier -- what the rendezvous of the 8 waves costs)
    asm volatile("" ::: "memory");
  };
  // relu-on-load: applied ONCE per element, in LDS, by the wave that requested the copy (the transform lanes would apply it
  // 2.25 times per element); read and write halves apart, a quad of MFMAs in between.  Out-of-image slots hold zeros: unchanged.
  f32x4 rl[3];
  auto relu_read = [&](int pb, int half) __attribute__((always_inline)) {       // half 0: pieces 0-2, half 1: pieces 3-4
    const char* b = sR + pb * (2 * F4_RAW_BYTES) + cw * 1024 + lane * 16 + half * 3 * 4096;
#pragma unroll
    for (int j = 0; j < 3; ++j) if (half == 0 || j < 2) rl[j] = *reinterpret_cast<const f32x4*>(b + j * 4096);
  };
  auto relu_write = [&](int pb, int half) __attribute__((always_inline)) {
    char* b = sR + pb * (2 * F4_RAW_BYTES) + cw * 1024 + lane * 16 + half * 3 * 4096;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (half == 1 && j == 2) break;
      f32x4 f = rl[j];
      asm("v_max_f32 %0, 0, %0" : "+v"(f.x)); asm("v_max_f32 %0, 0, %0" : "+v"(f.y));
      asm("v_max_f32 %0, 0, %0" : "+v"(f.z)); asm("v_max_f32 %0, 0, %0" : "+v"(f.w));
      *reinterpret_cast<f32x4*>(b + j * 4096) = f;
    }
  };

  // UPS: staged piece j (0; 1: waves 4-5 only) of pair pc into staging buffer lb
  const unsigned l_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sL;
  auto copy_l1 = [&](int pc, int lb, int j) __attribute__((always_inline)) {
    const unsigned so = (unsigned)pc * 32u;
    const unsigned lds = l_lds0 + (unsigned)lb * (unsigned)F4_L_BYTES + (unsigned)(j == 0 ? cw : 4 + cw) * 1024u;
    const unsigned o = j == 0 ? lro[0] : lro[1];
    FISR_F4_DMA1(rs0, o, so, lds);
  };
  // UPS: quad q = 64 cw + lane of the 9 x 17 quads of a halo tile (waves 4-6; the lanes behind quad 152 repeat it): staging
  // pixels (qy, qx) .. (qy + 1, qx + 1) -> halo pixels (2 qy + a, 2 qx + b); one 16-byte plane entry per pixel and chunk in, the
  // chunk's half of four raw records out (record slot and half as the raw copies of the other instantiations lay them out).
  int e_la = 0, e_r0 = 0, e_r1 = 0;
  if constexpr (UPS) {
    const int q = min(cw * 64 + lane, 152), qy = q / 17, qx = q - qy * 17;
    e_la = (qy * F4_L_COLS + qx) * 16;
    const int f16 = ((qy >> 1) & 1) * 16;
    e_r0 = (2 * qy * F4_HW + ((qx & 1) ? 18 : 0) + (qx >> 1)) * 32 + f16;
    e_r1 = (2 * qy * F4_HW + ((qx & 1) ? 26 : 9) + (qx >> 1)) * 32 + f16;
  }
  f32x4 e_tl, e_tr, e_bl, e_br;
  auto ex_read = [&](int lb, int c) __attribute__((always_inline)) {
    const char* b = sL + lb * F4_L_BYTES + c * F4_L_PLANE + e_la;
    e_tl = *reinterpret_cast<const f32x4*>(b);
    e_tr = *reinterpret_cast<const f32x4*>(b + 16);
    e_bl = *reinterpret_cast<const f32x4*>(b + F4_L_COLS * 16);
    e_br = *reinterpret_cast<const f32x4*>(b + F4_L_COLS * 16 + 16);
  };
  // (ey0, ex0: origin of the item the pair belongs to)
  auto ex_write = [&](int pb, int c, int ey0, int ex0) __attribute__((always_inline)) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    // a * 0.5 + c: the inline constant sits in the low half of its 64-bit operand, op_sel_hi = 0 hands it to both halves
    auto fma2 = [](f2_t a, f2_t c2) { f2_t r; asm("v_pk_fma_f32 %0, %1, 0.5, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c2)); return r; };
    auto sub2 = [](f2_t a, f2_t b2) { f2_t r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b2)); return r; };
    f32x4 oD, oC, oB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f2_t tl = {e_tl[2 * h], e_tl[2 * h + 1]}, tr = {e_tr[2 * h], e_tr[2 * h + 1]};
      const f2_t bl = {e_bl[2 * h], e_bl[2 * h + 1]}, br = {e_br[2 * h], e_br[2 * h + 1]};
      const f2_t A = fma2(sub2(tr, tl), tl);        // odd row, odd column: top
      const f2_t B = fma2(sub2(br, bl), bl);        //                      bottom = even row, odd column
      const f2_t C = fma2(sub2(br, tr), tr);        // odd row, even column
      const f2_t D = fma2(sub2(B, A), A);
      oD[2 * h] = D.x; oD[2 * h + 1] = D.y; oC[2 * h] = C.x; oC[2 * h + 1] = C.y; oB[2 * h] = B.x; oB[2 * h + 1] = B.y;
    }
    f32x4 oE = e_br;                                    // even row, even column
    const bool interior = ey0 > 0 && ex0 > 0 && ey0 + F4_TH + 1 <= p.H && ex0 + F4_TW + 1 <= p.W;     // (uniform) the whole halo inside the map
    if (!interior) {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int q = min(cw * 64 + l, 152), qy = q / 17, qx = q - qy * 17;
      const int gy = ey0 - 1 + 2 * qy, gx = ex0 - 1 + 2 * qx;
      const bool ya = gy >= 0 && gy < p.H, yb = gy + 1 < p.H, xa = gx >= 0 && gx < p.W, xb = gx + 1 < p.W;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      oD = ya && xa ? oD : z; oC = ya && xb ? oC : z; oB = yb && xa ? oB : z; oE = yb && xb ? oE : z;
    }
    char* r = sR + pb * (2 * F4_RAW_BYTES);
    const int a0 = e_r0 ^ (c * 16), a1 = e_r1 ^ (c * 16);
    *reinterpret_cast<f32x4*>(r + a0) = oD;
    *reinterpret_cast<f32x4*>(r + a1) = oC;
    *reinterpret_cast<f32x4*>(r + a0 + F4_HW * 32) = oB;
    *reinterpret_cast<f32x4*>(r + a1 + F4_HW * 32) = oE;
  };

  // (Measured and dropped: touching the residual tile's 128-byte lines ahead of the epilogue -- one dword per line into a register
  //  nobody reads, all 1024 lines two iterations before the end or 256 per iteration over the last four -- so that the epilogue's
  //  loads would hit L2.  The epilogue of a residual layer did shrink by 4k cycles, but the K loop grew by 10-15k: a CU gets ~10
  //  bytes per cycle out of HBM (its outstanding misses over the HBM latency) wherever the 128 KB are requested.  Third attempt,
  //  with the suspicion that the in-order vmcnt waits of the issuing waves were to blame: the touches as LDS-DMA copies into a dump
  //  area from transform waves that have NO other memory instruction (see UALL), in one burst 4 or 6 iterations before the end or a
  //  quarter per iteration from 8 or 12 before: epilogue -6k cycles, K loop +4..8k -- no wait involved, it is the CU's memory path.)

  // =========================== transform side (waves 0-3: tile half wave & 1, transform rows 3 (wave >> 1) ..) ===========================
  // lane -> (channel of the chunk, tile): 8 tiles of one tile row x 4 channels per ds_read_b32 phase.  The two waves of a tile
  // half share the work by OUTPUT rows: each reads the whole 6 x 6 patch, runs the vertical pass for its three rows and the
  // horizontal pass on them.  Everything is packed fp32 (v_pk_fma_f32 / v_pk_add_f32 on register pairs): the fp32 MFMA and the vector
  // ALU of a SIMD do NOT overlap (probed: a v_fma wave beside an MFMA wave on one SIMD takes the SUM of their times), so every vector
  // instruction of the K loop is paid in full on top of the 2304 MFMA cycles of a chunk.  Vertical pass: element-wise on column
  // pairs (6 instructions per pair and wave).  Horizontal pass: the op_sel bits feed one packed instruction from the halves of
  // different pairs, so the whole 6 -> 6 transform of a row is 6 instructions,
  //     (a, c) = x2 * (-4, -1) + x4      (b, e) = x1 * (-4, -1) + x3      (t1, t3) = (b, e) * (1, 2) + (a, c)
  //     (t2, t4) = (a, c) - (b, e) * (1, 2)        (t0, t5) = (x0, x1) * 4 + ((x2, x3) * -5 + (x4, x5))
  // and leaves its outputs in the order t0 t5 t1 t3 t2 t4: that IS the order of the six positions of a row in V, U and the
  // accumulators (F4_PERM), so nothing is shuffled.  36 packed instructions per wave and chunk (scalar: 78).
  // lane -> (channel, tile): a ds_read_b32 phase (32 lanes) = 4 tiles of one tile row and the 4 tiles below them x 4 channels.
  // The two tile rows read opposite halves of their pixels' 32-byte records (f = bit 2 of the halo row), so the phase covers
  // 8 (slot mod 4, half) combinations x 4 channels = 32 banks.  A lane's patch rows 0-3 lie in halo rows of one f, rows 4-5 in
  // the other: two base addresses, which trade places with the chunk's parity.
  const int t_ch = lane & 3, t_t8 = (lane >> 2) & 7;
  const int t_tx = (lane >> 5) * 4 + (t_t8 & 3), t_tyl = t_t8 >> 2;
  const int t_ty = 2 * (wave & 1) + t_tyl, t_t16 = t_tyl * 8 + t_tx;
  const int t_ra = ((4 * t_ty) * F4_HW + t_tx) * 32 + (t_ty & 1) * 16 + t_ch * 4;      // chunk 0 of a pair: patch rows 0-3 | chunk 1: rows 4-5
  const int t_rb = t_ra ^ 16;                                                          // chunk 0 of a pair: patch rows 4-5 | chunk 1: rows 0-3
  const int t_voff = (wave & 1) * 1024 + (t_ch * 16 + (t_t16 ^ (t_ch << 1))) * 16;
  const f32x2 K8 = {8.f, 8.f}, K4 = {4.f, 4.f}, KM4 = {-4.f, -4.f}, KM5 = {-5.f, -5.f}, K2 = {2.f, 2.f}, K41 = {-4.f, -1.f}, K12 = {1.f, 2.f};
  auto pk_fma = [](f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; };
  auto pk_fnma = [](f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; };
  auto pk_add = [](f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
  auto pk_sub = [](f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; };
  f32x2 dp[6][3];                                  // the 6 x 6 patch of this lane's (tile, channel) as column pairs
  f32x2 zp[3][3];                                  // this wave's three rows: vertical pass, then horizontal pass in place
  auto tr_read = [&](int pb, int sub) __attribute__((always_inline)) {       // chunk `sub` of the pair in buffer pb
    const char* r03 = sR + pb * (2 * F4_RAW_BYTES) + (sub ? t_rb : t_ra);
    const char* r45 = sR + pb * (2 * F4_RAW_BYTES) + (sub ? t_ra : t_rb);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int cb = (j & 3) == 0 ? 0 : (j & 3) == 1 ? 9 : (j & 3) == 2 ? 18 : 26;     // column 4 tx + j sits at cb + tx + j / 4
        dp[r][j >> 1][j & 1] = *reinterpret_cast<const float*>((r < 4 ? r03 : r45) + (r * F4_HW + cb + (j >> 2)) * 32);
      }
  };
  auto tr_col = [&](auto rh_tag, int c) __attribute__((always_inline)) {          // rows 3 rh .. 3 rh + 2 of B^T applied down the column pair c
    constexpr int RH = decltype(rh_tag)::value;
    if constexpr (RH == 0) {
      const f32x2 a = pk_fma(dp[2][c], KM4, dp[4][c]), b = pk_fma(dp[1][c], KM4, dp[3][c]);
      zp[0][c] = pk_fma(dp[0][c], K4, pk_fma(dp[2][c], KM5, dp[4][c]));
      zp[1][c] = pk_add(a, b);
      zp[2][c] = pk_sub(a, b);
    } else {
      const f32x2 cc = pk_sub(dp[4][c], dp[2][c]), e = pk_sub(dp[3][c], dp[1][c]);
      zp[0][c] = pk_fma(e, K2, cc);
      zp[1][c] = pk_fnma(e, K2, cc);
      zp[2][c] = pk_fma(dp[1][c], K4, pk_fma(dp[3][c], KM5, dp[5][c]));
    }
  };
  auto tr_row = [&](int i) __attribute__((always_inline)) {                       // -> (t0, t5), (t1, t3), (t2, t4)
    const f32x2 p01 = zp[i][0], p23 = zp[i][1], p45 = zp[i][2];
    f32x2 ac, be, t05, t13, t24;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(ac) : "v"(p23), "s"(K41), "v"(p45));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(be) : "v"(p01), "s"(K41), "v"(p23));
    t13 = pk_fma(be, K12, ac);
    t24 = pk_fnma(be, K12, ac);
    t05 = pk_fma(p01, K4, pk_fma(p23, KM5, p45));
    zp[i][0] = t05; zp[i][1] = t13; zp[i][2] = t24;
  };
  // the 18 slots 18 rh + 6 i + .. of this wave: four position quads and half of quad 4
  auto quad_of = [](f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); };
  // (SHARE: vchunk = the chunk's index in its item -- the same 16 / 8 bytes go to that chunk's image in p.vscr)
  auto tr_write = [&](auto rh_tag, int vbuf, int i, int vchunk = 0) __attribute__((always_inline)) {     // what row i (0..2) of this wave completes
    constexpr int RH = decltype(rh_tag)::value;
    char* vb = sV + vbuf * F4_V_BYTES + t_voff;
    const unsigned gso = (unsigned)((FISR_F4ABL & 32768) ? vchunk & 3 : vchunk) * (unsigned)F4_V_BYTES;
    auto put4 = [&](int off, f32x4 v) __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(vb + off) = v;
      if constexpr (SHARE && !(FISR_F4ABL & 8192)) wf4_store16<FISR_F4_VST_AUX>(__builtin_bit_cast(wf4_u32x4, v), rsv, (unsigned)t_voff, gso + (unsigned)off);
    };
    auto put2 = [&](int off, f32x2 v) __attribute__((always_inline)) {
      *reinterpret_cast<f32x2*>(vb + off) = v;
      if constexpr (SHARE && !(FISR_F4ABL & 8192)) wf4_store8<FISR_F4_VST_AUX>(__builtin_bit_cast(wf4_u32x2, v), rsv, (unsigned)t_voff, gso + (unsigned)off);
    };
    if constexpr (RH == 0) {
      if (i == 0) put4(0, quad_of(zp[0][0], zp[0][1]));
      if (i == 1) {
        put4(2048, quad_of(zp[0][2], zp[1][0]));
        put4(2 * 2048, quad_of(zp[1][1], zp[1][2]));
      }
      if (i == 2) {
        put4(3 * 2048, quad_of(zp[2][0], zp[2][1]));
        put2(4 * 2048, zp[2][2]);
      }
    } else {
      if (i == 0) {
        put2(4 * 2048 + 8, zp[0][0]);
        put4(5 * 2048, quad_of(zp[0][1], zp[0][2]));
      }
      if (i == 1) put4(6 * 2048, quad_of(zp[1][0], zp[1][1]));
      if (i == 2) {
        put4(7 * 2048, quad_of(zp[1][2], zp[2][0]));
        put4(8 * 2048, quad_of(zp[2][1], zp[2][2]));
      }
    }
  };
  typedef std::integral_constant<int, 0> rh0_t;
  typedef std::integral_constant<int, 1> rh1_t;
  auto transform = [&](auto rh_tag, int pb, int sub, int vbuf) __attribute__((always_inline)) {       // (prologue: the whole transform at once)
    tr_read(pb, sub);
#pragma unroll
    for (int c = 0; c < 3; ++c) tr_col(rh_tag, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) { tr_row(i); tr_write(rh_tag, vbuf, i); }
  };

  // =========================== MFMA side (all waves) ===========================
  // 16-channel quarter, 16-tile half.  GENERAL: the quarters of the second tile half are rotated by two, so that the two waves of a
  // SIMD (w and w + 4) own quarters {q, q ^ 2} -- in a layer with at most 32 output channels (one N block, half of it padding:
  // PWC-Net's last dense convolutions and dc_conv6) every SIMD then hosts exactly ONE wave with work for the matrix pipe.
  const int th = wave >> 2, cq = (GENERAL && FISR_F4_PADSKIP) ? ((wave & 3) ^ (th << 1)) : (wave & 3);
  // GENERAL: a wave ALL of whose items lie in the padding of the 64-channel block (zero weights, nothing stored) runs the item loop
  // without fragment reads, MFMAs and epilogue -- it keeps its copy / transform duties and the barriers.  The fp32 MFMA holds its
  // wave's issue slot, so the lone MFMA wave of a SIMD runs its 36 MFMAs per chunk without a partner to wait for.  (Decided once
  // per wave, outside the item loop: the two loops share no live accumulators.  A per-item decision -- the second N block of a
  // 96-channel layer -- was built first: the accumulators became values merged at every branch, 170-380 spilled registers.)
  // (Tried as well: the second N block of a 96-channel layer -- with gridDim.x / 8 a multiple of the N blocks all items of a
  //  workgroup lie in one block, so the decision is still per wave.  Same-box A/B: 1.6 ms per stack SLOWER -- the items are dealt
  //  round-robin, so the workgroups of the half-empty block only finish early, and running ahead of their neighbours they no longer
  //  share the input tile with them in L2.)
  const bool pad_wave = GENERAL && FISR_F4_PADSKIP && p.CoutPad == F4_BN && cq * 16 >= p.Cout;
  constexpr bool MMA_HERE = true;                  // (shadowed inside run_items by its mma_tag)
  const char* const fu = sU + cq * 1024 + lane * 16;
  const char* const fv = sV + th * 1024 + ((lane & 0x30) | ((lane & 15) ^ ((lane >> 4) << 1))) * 16;
  f32x4 acc[36];
  // The fragments of a position quad are requested two quads (8 MFMAs = 256 cycles of the pipe) ahead, and the LAST quad of a
  // chunk is multiplied behind the barrier, in the next iteration, from registers loaded before it: a wave issues in order, so
  // without the skew every iteration opens with both waves of a SIMD waiting for an LDS round trip and the matrix pipe idle.
  // (First iteration of an item: nothing is pending -- the accumulators of the last quad are zeroed instead -- and the other
  // eight quads start from C = 0, an inline constant: the 144 accumulators are never zeroed by hand.  The last chunk's quad is
  // flushed behind the loop.)
  f32x4 ra[3], rb[3];                              // ring of three quads; slot 2 holds the pending last quad across the barrier
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bias4 = zero4;                             // bias of this lane's four channels (of the current item)
  static_assert(wf4_slot(1, 1) == 8, "the bias rides in accumulator 8 (quad 2, element 0)");
#define FISR_F4_MMA4C(Q, A, B, C0_, C1_, C2_, C3_)                                                  \
  if constexpr (MMA_HERE && !(FISR_F4ABL & 16)) {                                                     \
  acc[4 * (Q)]     = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B).x, C0_, 0, 0, 0);              \
  acc[4 * (Q) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B).y, C1_, 0, 0, 0);              \
  acc[4 * (Q) + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B).z, C2_, 0, 0, 0);              \
  acc[4 * (Q) + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B).w, C3_, 0, 0, 0); }
#define FISR_F4_MMA4(Q, A, B) FISR_F4_MMA4C(Q, A, B, acc[4 * (Q)], acc[4 * (Q) + 1], acc[4 * (Q) + 2], acc[4 * (Q) + 3])
// first chunk of an item: C = 0 -- except position (1, 1) (slot 8), which starts from the BIAS: A^T has a 1 in column 1 of every
// row, so M[1][1] enters all 16 outputs of a tile with coefficient 1 and the bias add of the epilogue costs nothing.
#define FISR_F4_MMA4Z(Q, A, B) FISR_F4_MMA4C(Q, A, B, ((Q) == 2 ? bias4 : zero4), zero4, zero4, zero4)

  // ---- prologue of the workgroup's FIRST item: raw(0), U(0), raw(1), raw(2), raw(3) requested; raw(0) -> V[0] ----
  if (UPS && wave >= 4) {        // UPS: staged pairs 0 and 1, U(0); everything landed before the barrier (the quads read other waves' pieces)
    copy_l1(0, 0, 0);
    if (cw < 2) copy_l1(0, 0, 1);
#pragma unroll
    for (int j = 0; j < (UALL ? 9 : 5); ++j) copy_u1(cur.nblk, 0, 0, j);
    copy_l1(1, 1, 0);
    if (cw < 2) copy_l1(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave >= 4) {
    raw_offsets(true);
#pragma unroll
    for (int j = 0; j < 5; ++j) copy_pair1(0, 0, j);
#pragma unroll
    for (int j = 0; j < (UALL ? 9 : 5); ++j) copy_u1(cur.nblk, 0, 0, j);
    if (2 >= nch0) raw_offsets(false);
#pragma unroll
    for (int j = 0; j < 5; ++j) copy_pair1(1, 1, j);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // pair 0, U(0) landed; pair 1 in flight
    if constexpr (RELU_IN) { relu_read(0, 0); relu_write(0, 0); relu_read(0, 1); relu_write(0, 1); }
  } else if (!UALL) {
#pragma unroll
    for (int j = 0; j < 4; ++j) copy_u1(cur.nblk, 0, 0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  lds_barrier();
  if constexpr (UPS) {
    if (wave >= 4 && cw < 3) { ex_read(0, 0); ex_write(0, 0, cur.y0, cur.x0); ex_read(0, 1); ex_write(0, 1, cur.y0, cur.x0); }
    lds_barrier();
  }
  if (wave < 2) transform(rh0_t{}, 0, 0, 0);
  else if (wave < 4) transform(rh1_t{}, 0, 0, 0);
  lds_barrier();
  if (FISR_F4_TRACE && p.trace) t_first = __builtin_readcyclecounter();

  // ---- K loop of one item ----
  // g: the workgroup's running chunk counter (items have an even number of chunks, so its parity is k's).  Iteration g multiplies
  // chunk g and transforms chunk g+1 out of its raw pair.  Raw pairs: two buffers; pair m+1 is requested in the odd iteration
  // 2m-1 (pair m-1 was read for the last time in 2m-2), waited for and relu'd late in the even iteration 2m, read from 2m+1 on:
  // it has 1.6 iterations to come from HBM.  Weight copies early in the iteration (they have the rest of it to land), one behind
  // each quad of MFMAs; the transform between the quads that follow its LDS reads.  Behind an item's last chunk the streams
  // continue with the next item's first ones (behind the last item they repeat a chunk: the copy COUNT per iteration stays fixed
  // for the counted waits).
  int par = 0;                                     // V / U buffer of the chunk about to be multiplied
  int pbt = 0;                                     // raw pair buffer that holds chunk g+1
  typedef std::integral_constant<bool, true> first_t;
  typedef std::integral_constant<bool, false> rest_t;
  typedef std::integral_constant<bool, true> odd_t;
  typedef std::integral_constant<bool, false> even_t;
  auto k_iter = [&](auto role_tag, auto first_tag, auto odd_tag, auto mma_tag, auto fl_tag, int k) __attribute__((always_inline)) {
    constexpr int FL = decltype(fl_tag)::value;            // SHARE: bit 0 tr_on, bit 1 vc_on (transform waves), bit 2 rp_on (copy waves); else 7
    constexpr bool MMA_HERE = decltype(mma_tag)::value;    // false: a padding wave's iteration (GENERAL) -- copies, transform, barrier only
    constexpr int ROLE = decltype(role_tag)::value;        // 0 / 1: transform wave of rows 0-2 / 3-5; 2: copy wave
    constexpr bool FIRST = decltype(first_tag)::value;     // first chunk of an item
    constexpr bool ODD = decltype(odd_tag)::value;         // k odd
    typedef typename std::conditional<ROLE == 1, rh1_t, rh0_t>::type RH;
    const int buf = par;
    const char* ub = fu + buf * F4_U_BYTES;
    // SHARE: what this iteration's streams do is decided by the item their target chunk belongs to.
    //   tr_on  chunk k + 1 is a producer's: transformed here (and stored to p.vscr)
    //   vc_on  chunk k + 2 is a consumer's: copied from p.vscr into ring slot k & 3 (= (k + 2 + 2) & 3)
    //   rp_on  the raw stream (pair (k + 3) / 2 requested in odd k, relu'd in the even k behind it) feeds a producer
    // Transitions, with nch % 4 == 0 (pair buffers: pair m in RAW[m & 1]; P chunk c in V[c & 1]; C chunk c in ring slot (c + 2) & 3):
    //   P -> C  chunk 0 / 1 of C go to RAW[0] / RAW[1] in k = nch - 2 / nch - 1: P's pair np - 2 (RAW[0]) was read for the last time
    //           in k = nch - 4, pair np - 1 (RAW[1]) in k = nch - 2; chunks 2 / 3 go to V[0] / V[1] in C's k = 0 / 1, behind P's chunks
    //           nch - 2 / nch - 1 (the pending last quad of a chunk lives in registers).
    //   C -> P  pair 0 / 1 of P are requested into RAW[0] / RAW[1] in k = nch - 3 / nch - 1 -- ring slots 2 / 3, i.e. chunks
    //           nch - 4 / nch - 3, both multiplied by then; chunk 0 of P is transformed into V[0] in k = nch - 1 while C's chunk
    //           nch - 1 is read from slot 1; no V copy is issued in k >= nch - 2.
    // (compile-time per iteration body -- FL, chosen by k_loop's dispatch: as run-time branches inside the body the flags cost
    //  250-390 spilled registers)
    const bool cur_p = SHARE ? j_cur == 0 : true;
    constexpr bool tr_on = !SHARE || (FL & 1), vc_on = SHARE && (FL & 2), rp_on = !SHARE || (FL & 4);
    const char* vb = fv + (SHARE && !cur_p ? ring_off((k + 2) & 3) : buf * F4_V_BYTES);
    // U(k+1): of this item, of the next item (chunk 0), or a repeated chunk behind the last item
    const bool u_here = k + 1 < nch;
    const int ku = u_here ? k + 1 : (has_next ? 0 : nch - 1);
    const int nblk_cur = cur.nblk, nblk_nxt = nxt.nblk;       // (values first: a conditional between the captured structs' fields is a
    const int u_nblk = u_here || !has_next ? nblk_cur : nblk_nxt;     //  select of ADDRESSES into the closure, which then cannot be promoted to registers)
    if constexpr (MMA_HERE) {
      ra[0] = *reinterpret_cast<const f32x4*>(ub);
      rb[0] = *reinterpret_cast<const f32x4*>(vb);
      ra[1] = *reinterpret_cast<const f32x4*>(ub + 4096);
      rb[1] = *reinterpret_cast<const f32x4*>(vb + 2048);
    }
    if (ROLE < 2 && !(FISR_F4ABL & 4) && tr_on) tr_read(pbt, ODD ? 0 : 1);
    // odd iterations request the raw pair (k + 3) / 2: of this item, of the next one (its geometry from k = nch - 3 on), or a repeat
    int pc = 0;
    int ey0 = 0, ex0 = 0;
    if constexpr (ROLE == 2 && UPS && !ODD) {
      // UPS: even iterations blend the pair (k + 2) / 2 out of staging buffer pbt ^ 1 and request the pair (k + 4) / 2 into the other
      const int pp = (k + 4) >> 1, np = nch >> 1;
      pc = pp < np ? pp : (has_next ? pp - np : np - 1);
      if (k == nch - 4 && has_next) ups_geom(nxt);
      const int cy0 = cur.y0, cx0 = cur.x0, ny0 = nxt.y0, nx0 = nxt.x0;
      const bool of_next = k + 2 >= nch;
      ey0 = of_next ? ny0 : cy0; ex0 = of_next ? nx0 : cx0;
    }
    if (ROLE == 2 && ODD && !UPS && rp_on) {
      const int pp = (k + 3) >> 1, np = nch >> 1;
      pc = pp < np ? pp : (has_next ? pp - np : np - 1);
      const bool rfirst = 2 * pc < nch0;
      bool redo = rfirst != ro_first;
      if (k == nch - 3 && has_next) { raw_geom(nxt); redo = true; }
      if (redo) raw_offsets(rfirst);
    }
    if constexpr (FIRST) {
      if constexpr (MMA_HERE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[32 + r] = zero4;
      }
    } else {
      FISR_F4_MMA4(8, ra[2], rb[2])
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (MMA_HERE && q < 7) {
        ra[(q + 2) % 3] = *reinterpret_cast<const f32x4*>(ub + (q + 2) * 4096);
        rb[(q + 2) % 3] = *reinterpret_cast<const f32x4*>(vb + (q + 2) * 2048);
      }
      if constexpr (FIRST) { FISR_F4_MMA4Z(q, ra[q % 3], rb[q % 3]) } else { FISR_F4_MMA4(q, ra[q % 3], rb[q % 3]) }
      if constexpr (UALL) {
        if (ROLE == 2 && !(FISR_F4ABL & 1)) {
          if (q < 4) { copy_u1(u_nblk, ku, buf ^ 1, 2 * q); copy_u1(u_nblk, ku, buf ^ 1, 2 * q + 1); }
          if (q == 4) copy_u1(u_nblk, ku, buf ^ 1, 8);
        }
      } else if constexpr (RAWE) {
        // odd iterations: the transform waves issue all 36 weight copies (two behind quad 0, one behind each later quad), even
        // iterations: the copy waves
        if (!(FISR_F4ABL & 1) && ((ROLE < 2) == ODD)) {
          if (q == 0) { copy_u9(u_nblk, ku, buf ^ 1, 0); copy_u9(u_nblk, ku, buf ^ 1, 1); }
          else copy_u9(u_nblk, ku, buf ^ 1, q + 1);
        }
      } else if (!(FISR_F4ABL & 1) && q >= FISR_F4_UQ && q - FISR_F4_UQ < (ROLE == 2 ? 5 : 4)) copy_u1(u_nblk, ku, buf ^ 1, q - FISR_F4_UQ);
      if constexpr (ROLE < 2) {
        if (!(FISR_F4ABL & 4) && tr_on) {
          if (q == FISR_F4_COLQ) { tr_col(RH{}, 0); tr_col(RH{}, 1); tr_col(RH{}, 2); }
          if (q >= FISR_F4_ROWQ && q < FISR_F4_ROWQ + 3) { tr_row(q - FISR_F4_ROWQ); tr_write(RH{}, buf ^ 1, q - FISR_F4_ROWQ, k + 1 < nch ? k + 1 : 0); }
        }
        if constexpr (SHARE) {      // V(k + 2) of a consumer: behind the stores of this iteration (P's last but one has both)
          if (vc_on) {
            const int c2 = k + 2 < nch ? k + 2 : k + 2 - nch;
            if (q == 5) { copy_v1(c2, k & 3, 0); copy_v1(c2, k & 3, 1); }
            if (q == 6) { copy_v1(c2, k & 3, 2); copy_v1(c2, k & 3, 3); }
            if (q == 7 && ROLE == 0) copy_v1(c2, k & 3, 4);
          }
        }
      } else if constexpr (UPS) {
        if constexpr (!ODD) {
          if (cw < 3) {
            if (q == 4) ex_read(pbt ^ 1, 0);
            if (q == 5) ex_write(pbt ^ 1, 0, ey0, ex0);
            if (q == 6) ex_read(pbt ^ 1, 1);
            if (q == 7) ex_write(pbt ^ 1, 1, ey0, ex0);
          }
          if (q == 5) { copy_l1(pc, pbt, 0); if (cw < 2) copy_l1(pc, pbt, 1); }     // (behind this iteration's five weight copies)
        }
      } else if constexpr (ODD) {
        constexpr int Q0 = RAWE ? 0 : 5;            // (RAWEARLY: first thing in the iteration -- this wave has no weight copies in it)
        if (rp_on) {
          if (q == Q0) { copy_pair1(pc, pbt ^ 1, 0); copy_pair1(pc, pbt ^ 1, 1); }
          if (q == Q0 + 1) { copy_pair1(pc, pbt ^ 1, 2); copy_pair1(pc, pbt ^ 1, 3); }
          if (q == Q0 + 2) copy_pair1(pc, pbt ^ 1, 4);
        }
      } else if (RELU_IN && rp_on) {
        // the pair requested an iteration ago is older than this iteration's weight copies so far (five; RAWEARLY: six, quads 0-4)
        if (q == 4) {
          if constexpr (UALL) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
          else if constexpr (RAWE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          relu_read(pbt ^ 1, 0);
        }
        if (q == 5) relu_write(pbt ^ 1, 0);
        if (q == 6) relu_read(pbt ^ 1, 1);
        if (q == 7) relu_write(pbt ^ 1, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (FISR_F4ABL & 2048) { }                                                // (ablation 2048: no wait for the copies at the end of an iteration)
    else if (ROLE == 2 && UPS && !ODD) {                                           // U(g+1) landed; the staged pair stays in flight
      if (cw < 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else if (RAWE && ROLE == 2 && ODD) { }                                  // RAWEARLY: only the raw pair is in flight, and it stays
    else if (RAWE && ROLE < 2 && !ODD) { }                                    // RAWEARLY: no copies of this wave in an even iteration
    else if (SHARE && ROLE < 2) {
      // V(k + 1)'s copies (requested an iteration ago) landed, V(k + 2)'s -- the wave's last 5 / 4 memory instructions -- stay in
      // flight; without V copies in this iteration: the stores of this iteration's transform may stay in flight (the older ones
      // are through: a consumer reads them an item later at the earliest), otherwise everything is waited for.  (Loads return
      // in order among themselves; no wait here assumes an order between loads and stores.)
      if (vc_on) { if (ROLE == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else if (tr_on) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    else if (ROLE == 2 && ODD && !UPS && rp_on) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // U(g+1) landed; the raw pair stays in flight
    else if (ROLE < 2 && UALL) { }                                           // (nothing of this wave's to wait for)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // U(g+1) (and the raw pair of the iteration before) landed
    lds_barrier();
    if (!ODD) pbt ^= 1;
    par ^= 1;
  };
  typedef std::integral_constant<int, 0> role0_t;
  typedef std::integral_constant<int, 1> role1_t;
  typedef std::integral_constant<int, 2> role2_t;
  unsigned long long t2[6] = {0, 0, 0, 0, 0, 0};   // timeline of the workgroup's SECOND item (steady state), trace runs only
  int n_done = 0;
  typedef std::integral_constant<int, 7> fl_all_t;
  auto k_loop = [&](auto role_tag, auto mma_tag) __attribute__((always_inline)) {
    const bool tr2 = FISR_F4_TRACE && p.trace && n_done == 1;
    if (tr2) t2[0] = __builtin_readcyclecounter();
    if constexpr (!SHARE) {
      k_iter(role_tag, first_t{}, even_t{}, mma_tag, fl_all_t{}, 0);
      if (tr2) t2[1] = __builtin_readcyclecounter();
      k_iter(role_tag, rest_t{}, odd_t{}, mma_tag, fl_all_t{}, 1);
      if (tr2) t2[2] = __builtin_readcyclecounter();
      for (int k = 2; k < nch; k += 2) { k_iter(role_tag, rest_t{}, even_t{}, mma_tag, fl_all_t{}, k); k_iter(role_tag, rest_t{}, odd_t{}, mma_tag, fl_all_t{}, k + 1); }
    } else {
      // SHARE: the item's iterations are straight-line sequences of bodies with compile-time flags -- one sequence per kind of
      // item (P -> C, C -> C, C -> P, C -> nothing), picked ONCE per item like the wave's role (a choice per iteration, i.e. a
      // switch inside the loop, merges the 144 accumulators at every arm: 1800 spilled registers).  Iterations 0 .. nch - 4 run the
      // item's own kind of body (HF), the last three (T1 T2 T3) hand the streams over to the next item.
      constexpr int ROLE = decltype(role_tag)::value;
      auto seq = [&](auto hf, auto tl1, auto tl2, auto tl3) __attribute__((always_inline)) {
        k_iter(role_tag, first_t{}, even_t{}, mma_tag, hf, 0);
        if (tr2) t2[1] = __builtin_readcyclecounter();
        k_iter(role_tag, rest_t{}, odd_t{}, mma_tag, hf, 1);
        if (tr2) t2[2] = __builtin_readcyclecounter();
        for (int k = 2; k < nch - 4; k += 2) { k_iter(role_tag, rest_t{}, even_t{}, mma_tag, hf, k); k_iter(role_tag, rest_t{}, odd_t{}, mma_tag, hf, k + 1); }
        k_iter(role_tag, rest_t{}, even_t{}, mma_tag, hf, nch - 4);
        k_iter(role_tag, rest_t{}, odd_t{}, mma_tag, tl1, nch - 3);
        k_iter(role_tag, rest_t{}, even_t{}, mma_tag, tl2, nch - 2);
        k_iter(role_tag, rest_t{}, odd_t{}, mma_tag, tl3, nch - 1);
      };
      typedef std::integral_constant<int, 0> f0_t;
      typedef std::integral_constant<int, 1> f1_t;      // transform
      typedef std::integral_constant<int, 2> f2_t;      // V copy
      typedef std::integral_constant<int, 3> f3_t;      // both (a producer's last but one iteration)
      typedef std::integral_constant<int, 4> f4_t;      // raw stream
      const bool cur_p = j_cur == 0, nx_p = has_next && nxt_p, nx_c = has_next && !nxt_p;
      if constexpr (ROLE < 2) {
        if (cur_p) seq(f1_t{}, f1_t{}, f3_t{}, f2_t{});                 // P -> C  (share >= 2: a producer is followed by a consumer)
        else if (nx_c) seq(f2_t{}, f2_t{}, f2_t{}, f2_t{});             // C -> C
        else if (nx_p) seq(f2_t{}, f2_t{}, f0_t{}, f1_t{});             // C -> P
        else seq(f2_t{}, f2_t{}, f0_t{}, f0_t{});                       // C, the workgroup's last item
      } else {
        if (cur_p) seq(f4_t{}, f0_t{}, f0_t{}, f0_t{});
        else if (nx_p) seq(f0_t{}, f4_t{}, f4_t{}, f4_t{});
        else seq(f0_t{}, f0_t{}, f0_t{}, f0_t{});
      }
    }
    if (tr2) t2[3] = __builtin_readcyclecounter();
  };

  auto run_items = [&](auto mma_tag) __attribute__((always_inline)) {
  constexpr bool MMA_HERE = decltype(mma_tag)::value;
  for (;;) {                                       // ---- work items of this workgroup ----
    {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int c0 = cur.nblk * F4_BN + cq * 16 + 4 * (l >> 4);
      bias4 = *reinterpret_cast<const f32x4*>(p.bias + (c0 < p.Cout ? c0 : 0));
    }
    if (wave < 2) k_loop(role0_t{}, mma_tag);
    else if (wave < 4) k_loop(role1_t{}, mma_tag);
    else k_loop(role2_t{}, mma_tag);
    FISR_F4_MMA4(8, ra[2], rb[2])                  // the last chunk's pending quad
    if (FISR_F4_TRACE && p.trace && n_done == 0) t_main = __builtin_readcyclecounter();

    // ---- epilogue: Y = A^T M A in registers (the bias came in through accumulator (1,1)), lane transposition, (+ residual), relu,
    //      16-byte stores.  No LDS memory, no barrier: V / U / RAW already belong to the next item.
    //      The accumulators give lane (tile = l & 15, channel group = l >> 4) four channels of a pixel: the four lanes that make up
    //      a 64-byte record are 16 lanes apart, and the memory pipeline merges ADJACENT lanes only -- stored like that, every lane is
    //      its own 16-byte request (1024 per wave and item; the stores cost 11k cycles of a 72k-cycle 64 -> 64 item, the residual
    //      loads as much again).  So the outputs are transposed across the lanes first (ds_bpermute_b32: the LDS crossbar, no LDS
    //      memory): lane l takes tile l >> 2, channel group l & 3 -- four neighbouring lanes move one 64-byte record, and the residual
    //      is loaded in that layout directly.
    if constexpr (MMA_HERE) {
      int l = lane;
      asm volatile("" : "+v"(l));                  // (recomputed per item: see raw_geom)
      const int bp_addr = (((l & 3) << 4) | (l >> 2)) << 2;       // byte address of the SOURCE lane of the transposition
      const int e_t = th * 16 + (l >> 2);
      const int e_ty = e_t >> 3, e_tx = e_t & 7;
      const int c0 = cur.nblk * F4_BN + cq * 16 + 4 * (l & 3);
      const bool c_ok = c0 < p.Cout;
      const int cq_shift = p.d2s_shift;
      const unsigned sub = (unsigned)c0 >> cq_shift;
      const unsigned sA = p.d2s ? (unsigned)(4 * p.W) << cq_shift : (unsigned)p.W * (unsigned)rec_cs;
      const unsigned sB = p.d2s ? 2u << cq_shift : (unsigned)rec_cs;
      const unsigned vC = p.d2s ? ((((sub >> 1) * 2u * (unsigned)p.W + (sub & 1u)) << cq_shift) + ((unsigned)c0 & ((1u << cq_shift) - 1u))) : (unsigned)c0;
      const unsigned out_bytes = p.d2s ? ((unsigned)(4 * p.H * p.W) << cq_shift) * 4u : (unsigned)(p.H * p.W) * (unsigned)rec_cs * 4u - (unsigned)rec_co * 4u;
      const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
          (char*)p.out + (p.d2s ? ((size_t)cur.nb * 2 * p.H * 2 * p.W << cq_shift) * 4 : ((size_t)cur.nb * img_px * rec_cs + rec_co) * 4), 0, out_bytes, 0x00020000);
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      // Record (i, j) of the lane's 4 x 4 pixels sits at byte  vbase + 4 (i sA + j sB):  one lane offset + a UNIFORM offset that rides
      // in the scalar-offset operand of the buffer instruction -- no vector instruction per record (they are paid in full beside the
      // MFMAs' pipe: the epilogue is bound by its vector instruction count).  Lanes outside the image get an offset behind the
      // buffer's end (loads return zeros, stores are dropped); only a tile on the image's right / bottom edge needs that per record.
      const int oy0 = cur.y0 + 4 * e_ty, ox0 = cur.x0 + 4 * e_tx;         // (in the item's sub-image)
      const int hs = GENERAL ? (p.H - cur.ry + dil - 1) / dil : p.H, ws = GENERAL ? (p.W - cur.rx + dil - 1) / dil : p.W;     // the sub-image's size
      const unsigned vbase = c_ok ? ((unsigned)(oy0 * dil + cur.ry) * sA + (unsigned)(ox0 * dil + cur.rx) * sB + vC) * 4u : OOB;
      const bool interior = cur.y0 + F4_TH <= hs && cur.x0 + F4_TW <= ws;      // (uniform)
      unsigned off[4][4];
      if (!interior) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) off[i][j] = ((oy0 + i < hs) & (ox0 + j < ws)) ? vbase : OOB;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) off[i][j] = vbase;
      }
      const unsigned sA4 = sA * 4u * (unsigned)dil, sB4 = sB * 4u * (unsigned)dil;      // (neighbouring pixels of a sub-image are dil apart)
#define FISR_F4_SOFF(I, J) ((unsigned)(I) * sA4 + (unsigned)(J) * sB4)
      // output transform, packed on the channel pairs (r0, r1), (r2, r3) of every accumulator: rows first (W = M A, 6 x 4 per pair),
      // then columns (Y = A^T W).  The residual records are requested after the first half of the row pass -- from there on the
      // dead halves of the accumulators make room for them -- and arrive under the rest of the transform.
      auto at_pk = [&](f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2& y0, f32x2& y1, f32x2& y2, f32x2& y3) __attribute__((always_inline)) {
        const f32x2 s1 = pk_add(m1, m2), d1 = pk_sub(m1, m2), s2 = pk_add(m3, m4), d2 = pk_sub(m3, m4);
        y0 = pk_add(pk_add(m0, s1), s2);
        y1 = pk_fma(d2, K2, d1);
        y2 = pk_fma(s2, K4, s1);
        y3 = pk_add(pk_fma(d2, K8, d1), m5);
      };
      f32x2 yp[4][4][2];                           // [row][column][channel pair]
      if (FISR_F4_TRACE && p.trace && n_done == 1) t2[4] = __builtin_readcyclecounter();
      f32x4 res[4][4];
      auto half = [&](auto h_tag) __attribute__((always_inline)) {
        constexpr int h = decltype(h_tag)::value;
        f32x2 wp[6][4];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          f32x2 m[6];
#pragma unroll
          for (int b = 0; b < 6; ++b) m[b] = f32x2{acc[wf4_slot(a, b)][2 * h], acc[wf4_slot(a, b)][2 * h + 1]};
          at_pk(m[0], m[1], m[2], m[3], m[4], m[5], wp[a][0], wp[a][1], wp[a][2], wp[a][3]);
        }
        if constexpr (HAS_RES) {
          if (h == 0) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((char*)p.res + (size_t)cur.nb * img_px * p.Cout * 4, 0, out_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) res[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[i][j], FISR_F4_SOFF(i, j), 0));
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) at_pk(wp[0][j], wp[1][j], wp[2][j], wp[3][j], wp[4][j], wp[5][j], yp[0][j][h], yp[1][j][h], yp[2][j][h], yp[3][j][h]);
      };
      half(std::integral_constant<int, 0>{});
      half(std::integral_constant<int, 1>{});
      f32x4 pm[2][2];                              // 2x2 max pooling of the lane's 4 x 4 pixels (p.pool_out)
      // The stores' uniform offsets i sA4 + j sB4 are RUNNING scalar sums made opaque to the compiler: as sixteen loop-invariant
      // values it kept them in SGPRs across the K loops, spilled them into VGPR lanes and brought each back with a v_readlane
      // + the 4 wait states "VALU writes SGPR -> VMEM reads it" needs -- a vector instruction and 16 idle cycles per store in the
      // stage of the kernel that is bound by exactly that (r04).  A scalar add in front of the store has no such hazard.
      unsigned so_row = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned so_ij = so_row;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if FISR_F4_SOFF_RUN
          asm volatile("" : "+s"(so_ij));
#define FISR_F4_SOFF_ST(I, J) so_ij
#else
#define FISR_F4_SOFF_ST(I, J) FISR_F4_SOFF(I, J)
#endif
          const float y0 = yp[i][j][0][0], y1 = yp[i][j][0][1], y2 = yp[i][j][1][0], y3 = yp[i][j][1][1];
          f32x4 o;
          o[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y0)));
          o[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y1)));
          o[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y2)));
          o[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y3)));
          if constexpr (HAS_RES) o += res[i][j];
          if constexpr (GENERAL) {        // leaky relu: max(v, slope v), slope 1 when the layer has none
            const float sl = p.relu_out ? p.slope : 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t_;
              asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t_) : "s"(sl), "v"(o[e]));
              asm volatile("v_max_f32 %0, %1, %0" : "+v"(o[e]) : "v"(t_));
            }
          } else if (p.relu_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, 0, %0" : "+v"(o[e]));
          }
          // A 16-byte buffer store reads its data registers AFTER it has issued, and a vector instruction that overwrites one of
          // them in the very next slot wins that race now and then -- measured on gfx950 WITH a register in the soffset field, the
          // case the compiler's hazard recogniser exempts (createsVALUHazard): in the GENERAL instantiation the leaky relu's
          // temporary was allocated to dword 2 of the previous record and scheduled right behind its store, and pixel j came out
          // with channel 4g + 2 = pixel j + 1's channel 4g, in the younger waves of a SIMD only, in 0.02-0.15 % of the outputs,
          // differently on every run (r04; scripts/isa_store_hazard.py finds such pairs in a listing, tests/test_host.py runs it on
          // the built code object).  So every store has one wait state pinned behind it (wf4_store16).
          if (!(FISR_F4ABL & 512) || (i == 0 && j == 0)) {      // (ablation 512: one store per lane instead of 16)
            if (FISR_F4_STORE_AUX == 0) wf4_store16(__builtin_bit_cast(u32x4_t, o), os, off[i][j], FISR_F4_SOFF_ST(i, j));
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), os, off[i][j], FISR_F4_SOFF_ST(i, j), FISR_F4_STORE_AUX);
          }
          so_ij += sB4;
          if constexpr (POOL) {
            if ((i & 1) == 0 && (j & 1) == 0) pm[i >> 1][j >> 1] = o;
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) pm[i >> 1][j >> 1][e] = fmaxf(pm[i >> 1][j >> 1][e], o[e]);
            }
          }
        }
        so_row += sA4;
      }
#undef FISR_F4_SOFF_ST
      if constexpr (POOL) {       // ops.py:54 max_pool 2x2 / 2 of what was just stored: [N, H/2, W/2, Cout]
        const unsigned ph = (unsigned)p.H >> 1, pw = (unsigned)p.W >> 1;
        const unsigned pool_bytes = ph * pw * (unsigned)p.Cout * 4u;
        const __amdgpu_buffer_rsrc_t ps_ = __builtin_amdgcn_make_buffer_rsrc((char*)p.pool_out + (size_t)cur.nb * pool_bytes, 0, pool_bytes, 0x00020000);
        const unsigned pbase = c_ok ? (((unsigned)oy0 >> 1) * pw + ((unsigned)ox0 >> 1)) * (unsigned)p.Cout * 4u + (unsigned)c0 * 4u : OOB;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
          for (int bj = 0; bj < 2; ++bj) {
            const unsigned o_ = (interior || ((oy0 + 2 * bi < p.H) & (ox0 + 2 * bj < p.W))) ? pbase : OOB;
            if (FISR_F4_STORE_AUX == 0) wf4_store16(__builtin_bit_cast(u32x4_t, pm[bi][bj]), ps_, o_, ((unsigned)bi * pw + (unsigned)bj) * (unsigned)p.Cout * 4u);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pm[bi][bj]), ps_, o_,
                                                        ((unsigned)bi * pw + (unsigned)bj) * (unsigned)p.Cout * 4u, FISR_F4_STORE_AUX);
          }
      }
    }
    if (FISR_F4ABL & 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ablation: wait for the stores here
    if (FISR_F4_TRACE && p.trace && n_done == 0) t_end1 = __builtin_readcyclecounter();
    if (FISR_F4_TRACE && p.trace && n_done == 1) t2[5] = __builtin_readcyclecounter();
    ++n_done;
    if (!has_next) break;
    if constexpr (SHARE) {
      cur = nxt;
      if (nxt_p) { b_cur += gridDim.x; j_cur = 0; } else ++j_cur;
      if (j_cur + 1 < share) { has_next = true; nxt_p = false; nxt.nblk = cur.nblk + 1; }
      else {
        has_next = b_cur + (int)gridDim.x < n_units;
        nxt_p = true;
        nxt = has_next ? item_of(b_cur + gridDim.x) : cur;
      }
    } else {
      b_cur += gridDim.x;
      cur = nxt;
      has_next = b_cur + (int)gridDim.x < n_items;
      nxt = has_next ? item_of(b_cur + gridDim.x) : cur;
    }
  }
  };
  if constexpr (GENERAL) {
    if (pad_wave) run_items(std::integral_constant<bool, false>{});
    else run_items(std::integral_constant<bool, true>{});
  } else {
    run_items(std::integral_constant<bool, true>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the repeated copies behind the last item still write LDS)
#undef FISR_F4_SOFF
#undef FISR_F4_MMA4
#undef FISR_F4_MMA4Z
#undef FISR_F4_MMA4C
#undef FISR_F4_DMA1
  if (FISR_F4_TRACE && p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_main; tr[2] = __builtin_readcyclecounter();
    tr[3] = t_end1; tr[4] = t_first; tr[5] = t_real; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = (unsigned long long)n_done;
  }
  if (FISR_F4_TRACE && p.trace && (tid & 63) == 0 && n_done > 1) {   // second row block: per WAVE {item start, after iteration 0, 1, K loop end, epilogue start of the stores.., end}
    unsigned long long* tr = p.trace + ((size_t)gridDim.x + (size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) tr[i] = t2[i];
  }
}

#undef FISR_F4_BEGIN
#undef FISR_F4_COPY
#undef FISR_F4_COPYU
#undef FISR_F4_END

// ---- host side: weight slabs, eligibility, launch ----

// Winograd F(4x4,3x3) weights (conv3x3_wf4.h): U = G g G^T per (ci, co) with the 6x3 G of the interpolation points 0, +-1, +-2, inf,
// computed in double and rounded once to fp32, stored as the kernel's LDS image
// [Cin/4][CoutPad/64][slot quad 9][channel quarter 4][ci 4][channel 16][4 slots]  (slot of position (i, j): wf4_slot).
inline void pack_weights_wf4(const float* w, int ci, int co, int cin_pad, std::vector<char>& wp) {
  static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  const int nb = (co + F4_BN - 1) / F4_BN, nch = cin_pad / F4_CH;
  wp.assign((size_t)nch * nb * F4_U_BYTES, 0);
  for (int c = 0; c < ci; ++c)
    for (int n = 0; n < co; ++n) {
      double g[3][3], t[6][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)(a * 3 + b) * ci + c) * co + n];
      for (int i = 0; i < 6; ++i)
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
      const int kc = c / F4_CH, k = c % F4_CH, blk = n / F4_BN, q = (n % F4_BN) / 16, r = n % 16;
      float* slab = reinterpret_cast<float*>(wp.data() + ((size_t)kc * nb + blk) * F4_U_BYTES);
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          const int pos = wf4_slot(i, j);
          slab[(((pos >> 2) * 4 + q) * 64 + k * 16 + r) * 4 + (pos & 3)] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    }
}
// what conv3x3_wf4.h takes: whole 64-channel output blocks, whole 8-channel raw pairs per concat source, images addressed with
// 31-bit byte offsets (the out-of-range marker of its zero padding is 2^31)
inline bool wf4_fits(int h, int w, int c0, int c1, int co) {
  return co % F4_BN == 0 && c0 > 0 && c0 % (2 * F4_CH) == 0 && c1 % (2 * F4_CH) == 0 && (c0 + c1) / F4_CH >= 4 && (double)h * w * std::max(std::max(c0, c1), co) * 4.0 < 2147483648.0;
}
// the GENERAL instantiation: one source of whole raw pairs out of a buffer with pixel stride in_cs, an output range of a buffer
// with pixel stride out_cs (the 64-channel output blocks are padded: CoutPad), the same 31-bit byte offsets
inline bool wf4_fits_general(int h, int w, int c0, int in_cs, int out_cs) {
  return c0 > 0 && c0 % (2 * F4_CH) == 0 && c0 / F4_CH >= 4 && (double)h * w * std::max(in_cs, out_cs) * 4.0 < 2147483648.0;
}
// ... and where it is the faster of the two Winograd kernels (measured per map size and depth, scripts/conv_bench.py, 12 tiles:
// 68 x 124 maps 1.2-1.35x at every channel count, 34 x 62 maps 0.72-0.77x with 64-256 channels -- its 16 x 32-pixel items waste
// more of a small map than the 8 x 32 ones of conv3x3_wino8p.h -- but 1.04x / 1.4x on the 512-channel 34 x 62 / 17 x 31 maps,
// where the K loop is long enough to pay for the padding)
inline bool wf4_wins(int h, int w, int cin) { return (h >= 48 && w >= 64) || cin >= 512; }
// SHARE instantiations (r05): `share` consecutive N blocks of a tile per run, whole quadruples of chunks per item (the consumers' ring),
// one slice of V scratch per workgroup: wf4_vscr_bytes() for the largest grid the launcher uses
inline bool wf4_share_fits(int c0, int c1, int cout, int share) {
  const int nch = (c0 + c1) / F4_CH, nb = cout / F4_BN;
  return share >= 2 && nch % 4 == 0 && nch >= 8 && cout % F4_BN == 0 && nb % share == 0;
}
constexpr int F4_MAX_GRID = 256;
inline size_t wf4_vscr_bytes(int c0, int c1) { return (size_t)F4_MAX_GRID * ((c0 + c1) / F4_CH) * F4_V_BYTES; }

// The F(4x4,3x3) Winograd kernel (conv3x3_wf4.h; fp32, FISRnet's dense layers only; a.wpk = the conv's d_wu4).
inline hipError_t launch_conv_wf4(const ConvArgs& a, hipStream_t st) {
  static bool attr_done[64] = {};
  static int n_cu[64] = {};
  constexpr size_t lds = wf4_lds_bytes();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    const void* kerns[] = {reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, false>), reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, true>),
                           reinterpret_cast<const void*>(conv3x3_wf4_kernel<true, false>), reinterpret_cast<const void*>(conv3x3_wf4_kernel<true, true>),
                           reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, true, true>)};
    for (const void* k : kerns) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipError_t eu = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)wf4_lds_bytes_ups());
    if (eu != hipSuccess) return eu;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    attr_done[dev] = true;
  }
  const bool plain = a.in0_cs == a.C0 && (a.C1 == 0 || a.in1_cs == a.C1) && a.rec_cs == a.Cout && a.rec_co == 0 && a.slope == 0.f && a.dil == 1 &&
                     a.CoutPad == a.Cout;
  if (!plain) {
    // GENERAL (PWC-Net): channel ranges, padded output blocks, leaky relu, dilation -- one source, nothing fused
    if (a.C1 || a.in1 || a.res || a.d2s || a.relu_in || a.pool_out || a.ups || a.dil < 1 || a.CoutPad % F4_BN || a.Cout % 4 || a.Cout > a.CoutPad ||
        a.in0_cs < a.C0 || a.rec_co + a.Cout > a.rec_cs || ((a.in0_cs | a.rec_cs | a.rec_co) & 3) || !wf4_fits_general(a.H, a.W, a.C0, a.in0_cs, a.rec_cs))
      return hipErrorInvalidValue;
    static bool gattr[64] = {};
    if (!gattr[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, false, false, false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      gattr[dev] = true;
    }
    const int d = a.dil;
    const int gitems = (((a.W + d - 1) / d + F4_TW - 1) / F4_TW) * (((a.H + d - 1) / d + F4_TH - 1) / F4_TH) * d * d * a.N * (a.CoutPad / F4_BN);
    hipLaunchKernelGGL((conv3x3_wf4_kernel<false, false, false, false, true>), dim3(std::min(gitems, std::max(8, n_cu[dev] & ~7))), dim3(512), lds, st, a, gitems);
    return hipGetLastError();
  }
  if (!wf4_fits(a.H, a.W, a.C0, a.C1, a.Cout) || (a.res && a.d2s)) return hipErrorInvalidValue;
  // (fused pooling: instantiated for what FISRnet needs it for, the last conv of an encoder level -- residual, no relu-on-load)
  if (a.pool_out && (a.d2s || (a.H & 1) || (a.W & 1) || a.relu_in || !a.res)) return hipErrorInvalidValue;
  // (fused x2 bilinear: the decoder's resize convolution -- one source, no relu-on-load, no residual)
  if (a.ups && ((a.H & 1) || (a.W & 1) || a.C1 || a.relu_in || a.res || a.pool_out)) return hipErrorInvalidValue;
  const int items = ((a.W + F4_TW - 1) / F4_TW) * ((a.H + F4_TH - 1) / F4_TH) * a.N * (a.CoutPad / F4_BN);
  // one workgroup per CU (the kernel needs most of a CU's LDS and half its registers), a multiple of 8 so that the items of a
  // workgroup stay on one XCD
  const int grid = std::min(items, std::max(8, n_cu[dev] & ~7));
#ifndef FISR_F4_SHARE
  if (a.share >= 2) return hipErrorInvalidValue;      // (the SHARE instantiations are built into scripts/probes/wf4_bench only: see the kernel's header)
#else
  if (a.share >= 2) {
    // V sharing: runs of a.share N blocks; the grid counts runs
    if (!a.vscr || a.ups || !wf4_share_fits(a.C0, a.C1, a.Cout, a.share)) return hipErrorInvalidValue;
    static bool sattr[64] = {};
    if (!sattr[dev]) {
      const void* sk[] = {reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, false, false, false, false, true>),
                          reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, true, false, false, false, true>),
                          reinterpret_cast<const void*>(conv3x3_wf4_kernel<true, false, false, false, false, true>),
                          reinterpret_cast<const void*>(conv3x3_wf4_kernel<true, true, false, false, false, true>),
                          reinterpret_cast<const void*>(conv3x3_wf4_kernel<false, true, true, false, false, true>)};
      for (const void* k : sk) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
      }
      sattr[dev] = true;
    }
    const int sgrid = std::min(items / a.share, std::min(F4_MAX_GRID, std::max(8, n_cu[dev] & ~7)));
    if (a.pool_out) hipLaunchKernelGGL((conv3x3_wf4_kernel<false, true, true, false, false, true>), dim3(sgrid), dim3(512), lds, st, a, items);
    else if (a.relu_in) {
      if (a.res) hipLaunchKernelGGL((conv3x3_wf4_kernel<true, true, false, false, false, true>), dim3(sgrid), dim3(512), lds, st, a, items);
      else hipLaunchKernelGGL((conv3x3_wf4_kernel<true, false, false, false, false, true>), dim3(sgrid), dim3(512), lds, st, a, items);
    } else {
      if (a.res) hipLaunchKernelGGL((conv3x3_wf4_kernel<false, true, false, false, false, true>), dim3(sgrid), dim3(512), lds, st, a, items);
      else hipLaunchKernelGGL((conv3x3_wf4_kernel<false, false, false, false, false, true>), dim3(sgrid), dim3(512), lds, st, a, items);
    }
    return hipGetLastError();
  }
#endif
  if (a.ups) {
    hipLaunchKernelGGL((conv3x3_wf4_kernel<false, false, false, true>), dim3(grid), dim3(512), wf4_lds_bytes_ups(), st, a, items);
  } else if (a.pool_out) {
    hipLaunchKernelGGL((conv3x3_wf4_kernel<false, true, true>), dim3(grid), dim3(512), lds, st, a, items);
  } else if (a.relu_in) {
    if (a.res) hipLaunchKernelGGL((conv3x3_wf4_kernel<true, true>), dim3(grid), dim3(512), lds, st, a, items);
    else hipLaunchKernelGGL((conv3x3_wf4_kernel<true, false>), dim3(grid), dim3(512), lds, st, a, items);
  } else {
    if (a.res) hipLaunchKernelGGL((conv3x3_wf4_kernel<false, true>), dim3(grid), dim3(512), lds, st, a, items);
    else hipLaunchKernelGGL((conv3x3_wf4_kernel<false, false>), dim3(grid), dim3(512), lds, st, a, items);
  }
  return hipGetLastError();
}


}  // namespace fisr
