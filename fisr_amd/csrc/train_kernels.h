// Kernels of the training graph (FISRnet.py:175-497, ops.py:7-76 differentiated).  All fp32, NHWC.
//
//   train_pack_kernel      master weights (TF HWIO, device) -> the direct conv kernel's packed layout (conv3x3.h), either as
//                          they are (forward) or with the taps rotated by 180 degrees and Ci / Co swapped (data gradient:
//                          dx = conv(dy, rot180(w)^T), the same forward kernel then does the work)
//   train_wgrad_kernel     dW[tap][ci][co] += sum over pixels of x[p + tap][ci] * g[p][co]: a GEMM whose K is the pixel
//                          axis, on v_mfma_f32_32x32x2_f32 (A = 32 input channels x 2 pixels, B = 2 pixels x 32 output
//                          channels, one accumulator per tap), halo tile of x and tile of g staged in LDS
//   train_bgrad_kernel     db[co] += sum over pixels of g[p][co]
//   relu_bwd, maxpool2_bwd, upsample2_bwd (adjoint of the legacy bilinear), space_to_depth (adjoint of depth_to_space),
//   axpy, gather / scatter of channel ranges, the loss kernel (all seven terms of FISRnet.py:316-484 and their gradient
//   with respect to the twelve predicted frames of a level), Adam (tf.train.AdamOptimizer, TF 1.13)
#pragma once
#include "conv3x3.h"
#include "conv3x3_wino8p.h"

namespace fisr {

// ---- weights: HWIO -> packed rows of the direct fp32 kernel (pack_weights<float> in fisr_api.hip, on the device) ----
// transpose = 0: src[tap][c][n]; transpose = 1: the conv has ci' = co, co' = ci and src'[tap][c'][n'] = src[8 - tap][n'][c'].
__device__ __forceinline__ void train_pack_body(const float* __restrict__ w, int ci_src, int co_src, int transpose, int ci, int co,
                                                int cin_pad, int cout_pad, float* __restrict__ out, size_t first, size_t stride) {
  const size_t total = (size_t)(cin_pad / 16) * 9 * cout_pad * 16;
  for (size_t i = first; i < total; i += stride) {
    const int cc = (int)(i & 15);
    size_t t = i >> 4;
    const int row = (int)(t % cout_pad); t /= cout_pad;
    const int tap = (int)(t % 9);
    const int kc = (int)(t / 9);
    // row -> channel n (inverse of row = (n & ~31) + (wr & 3) + 8 * (wr >> 2) + 4 * wk, wi = n & 31, wk = wi >> 4, wr = wi & 15)
    int n;
    if (cout_pad == 16) n = row;
    else {
      const int r5 = row & 31, wk = (r5 >> 2) & 1, wr = (r5 & 3) + 4 * (r5 >> 3);
      n = (row & ~31) + 16 * wk + wr;
    }
    const int c = kc * 16 + cc;
    float v = 0.f;
    if (c < ci && n < co)
      v = transpose ? w[((size_t)(8 - tap) * ci_src + n) * co_src + c] : w[((size_t)tap * ci_src + c) * co_src + n];
    out[i] = v;
  }
}
__global__ void train_pack_kernel(const float* __restrict__ w, int ci_src, int co_src, int transpose, int ci, int co,
                                  int cin_pad, int cout_pad, float* __restrict__ out) {
  train_pack_body(w, ci_src, co_src, transpose, ci, co, cin_pad, cout_pad, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (size_t)gridDim.x * blockDim.x);
}

// ---- weights: HWIO -> Winograd slabs U = G g G^T of conv3x3_wino8p.h (pack_weights_wino in fisr_api.hip, on the device, fp32) ----
// One thread per (input channel c < cin_pad, output channel n < nb * 64); transpose as in train_pack_kernel.
__device__ __forceinline__ void train_pack_wino_body(const float* __restrict__ w, int ci_src, int co_src, int transpose, int ci, int co,
                                                     int cin_pad, int nb, char* __restrict__ out, size_t first, size_t stride) {
  // one thread per (output channel n, four input channels c .. c+3 = one 16-byte half record): 16 stores of 16 bytes
  const size_t total = (size_t)(cin_pad / 4) * nb * 64;
  for (size_t i = first; i < total; i += stride) {
    // lanes run along the source's contiguous axis: output channels of w[tap][c][n], input channels of the transposed
    // w[8 - tap][n][c] (lanes along n there read 16 bytes out of every 128-byte line they touch)
    const int n = transpose ? (int)(i / (cin_pad / 4)) : (int)(i % (nb * 64));
    const int c0 = 4 * (transpose ? (int)(i % (cin_pad / 4)) : (int)(i / (nb * 64)));
    f32x4 u[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      float g[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int tap = a * 3 + b;
          g[a][b] = (c < ci && n < co) ? (transpose ? w[((size_t)(8 - tap) * ci_src + n) * co_src + c] : w[((size_t)tap * ci_src + c) * co_src + n]) : 0.f;
        }
      float t[4][3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b]; t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]); t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]); t[3][b] = g[2][b];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        u[4 * a][e] = t[a][0]; u[4 * a + 1][e] = 0.5f * (t[a][0] + t[a][1] + t[a][2]); u[4 * a + 2][e] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        u[4 * a + 3][e] = t[a][2];
      }
    }
    const int kc = c0 / W_CH, h = (c0 % W_CH) >> 2;
    const int blk = n / W_BN, nl = n % W_BN;
    const int wi = nl & 31, wk = wi >> 4, wr = wi & 15;
    const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
    char* slab = out + ((size_t)kc * nb + blk) * W_SLAB;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
      *reinterpret_cast<f32x4*>(slab + ((size_t)pos * 64 + row) * W_REC + ((h ^ ((row >> 3) & 1)) * 16)) = u[pos];
  }
}
__global__ void train_pack_wino_kernel(const float* __restrict__ w, int ci_src, int co_src, int transpose, int ci, int co,
                                       int cin_pad, int nb, char* __restrict__ out) {
  train_pack_wino_body(w, ci_src, co_src, transpose, ci, co, cin_pad, nb, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                       (size_t)gridDim.x * blockDim.x);
}

// Every layout of every conv in ONE launch (the training step repacks all 138 convs after Adam: 640 launches of a few
// microseconds each otherwise).  blockIdx.y = 4 * conv + layout (0 direct, 1 direct transposed, 2 Winograd, 3 Winograd
// transposed); a null destination skips the layout.  PackDesc mirrors fisr_train_pack_desc (include/fisr.h).
struct PackDesc { const float* w; float* pk; float* pk_t; void* pkw; void* pkw_t; int ci, co; };
__global__ void train_pack_all_kernel(const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y >> 2];
  const int which = blockIdx.y & 3, transpose = which & 1;
  const int ci = transpose ? d.co : d.ci, co = transpose ? d.ci : d.co;
  const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  const int cin_pad = (ci + 15) / 16 * 16;
  if (which < 2) {
    float* out = transpose ? d.pk_t : d.pk;
    if (!out) return;
    const int nt = co < 16 ? 0 : (co <= 32 ? 1 : 2);       // nt_for<float> (fisr_api.hip)
    const int cout_pad = nt == 0 ? 16 : (co + 32 * nt - 1) / (32 * nt) * (32 * nt);
    train_pack_body(d.w, d.ci, d.co, transpose, ci, co, cin_pad, cout_pad, out, first, stride);
  } else {
    char* out = (char*)(transpose ? d.pkw_t : d.pkw);
    if (!out) return;
    train_pack_wino_body(d.w, d.ci, d.co, transpose, ci, co, cin_pad, (co + 15) / 16 * 16 / W_BN, out, first, stride);
  }
}

// ---- weight gradient ----
// FISR_GABL (diagnostics builds only, WRONG results): 1 tiles staged once only, 2 no MFMAs, 4 no final reduction / atomics
#ifndef FISR_GABL
#define FISR_GABL 0
#endif
struct WgradArgs {
  const float* x0; const float* x1; int C0, C1;     // conv input: cat(x0, x1) along channels (x1 nullable), pixel strides C0, C1
  const float* g;  int Cg;                          // gradient of the conv output [N,H,W,Cg] (dense)
  float* dw;       int ci, co;                      // HWIO [3][3][ci][co], accumulated (ci <= C0 + C1, co <= Cg)
  float* db;                                        // nullable: db[co] += sum over pixels of g (done by the first ci block)
  int N, H, W, relu_in, ksplit;
  unsigned long long* trace;                        // diagnostics builds: 8 words per workgroup (cycle counters), else null
};
// Tile geometry is a template parameter: the U-Net's maps go down to 3 x 3 pixels (level 1 at R/8 of a 24 x 24 patch), and a fixed
// 32-wide tile spends most of its MFMAs on padding there.  wgrad_tile() picks, per launch, the (TW, TH) with the least padding.
// TH is a multiple of 4 (one to four rows per wave); at most 128 pixels per tile keeps the register prefetch small enough for two
// workgroups per CU.
// LDS: the two 36 KB slots of the final reduction (the tiles, smaller, live in the same bytes before it) + the four waves' bias sums
constexpr size_t WG_RED_BYTES = (size_t)2 * 9 * 16 * 64 * 4;
constexpr size_t wgrad_lds_bytes() { return WG_RED_BYTES + 4 * 32 * 4; }
struct WgradTile { int tw, th; };
inline WgradTile wgrad_tile(int h, int w) {
  static const WgradTile cand[] = {{32, 4}, {16, 8}, {16, 4}, {8, 16}, {8, 8}, {8, 4}};   // larger tiles first: ties go to them
  WgradTile best = cand[0];
  double best_eff = -1.0;
  for (const WgradTile& c : cand) {
    const double eff = ((double)w / (((w + c.tw - 1) / c.tw) * c.tw)) * ((double)h / (((h + c.th - 1) / c.th) * c.th));
    if (eff > best_eff + 1e-9) { best_eff = eff; best = c; }
  }
  return best;
}

// Workgroup b runs on XCD b % 8 (observed; speed only).  The (ci block, co block) workgroups of one pixel-tile sequence read the
// same x and g tiles: give every XCD a contiguous range of the virtual ids so that they share that XCD's L2 instead of
// fetching the tiles once per XCD.
__device__ __forceinline__ int wg_xcd_order(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Register staging of one pixel tile of the weight-gradient kernels: halo tile of x (TH+2 x TW+2 pixels, 32 input channels from
// cib on, zero outside the image and behind the last channel) and tile of g (TH x TW pixels, 32 output channels from cob on).
template <int WG_TW, int WG_TH> struct WgStager {
  static constexpr int NX = ((WG_TH + 2) * (WG_TW + 2) * 8 + 255) / 256, NG = WG_TH * WG_TW * 8 / 256;
  f32x4 rx[NX], rg[NG];
  __device__ __forceinline__ void load(const WgradArgs& p, int tile, int tiles_x, int tiles_y, int cib, int cob, int tid) {
    int t = tile;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int nb = t / tiles_y;
    const int x0 = tx * WG_TW, y0 = ty * WG_TH;
#pragma unroll
    for (int k = 0; k < NX; ++k) {                   // x halo tile, 4 channels per thread
      const int i = tid + 256 * k;
      const int px = min(i >> 3, (WG_TH + 2) * (WG_TW + 2) - 1), q = i & 7;
      const int py = px / (WG_TW + 2), pxx = px - py * (WG_TW + 2);
      const int gy = y0 - 1 + py, gx = x0 - 1 + pxx, c = cib + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
        const size_t pix = ((size_t)nb * p.H + gy) * p.W + gx;
        if (c < p.C0) v = *reinterpret_cast<const f32x4*>(p.x0 + pix * p.C0 + c);            // (C0 % 4 == 0)
        else if (c < p.C0 + p.C1) v = *reinterpret_cast<const f32x4*>(p.x1 + pix * p.C1 + (c - p.C0));
      }
      rx[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {                   // g tile
      const int i = tid + 256 * k;
      const int px = i >> 3, q = i & 7;
      const int py = px / WG_TW, pxx = px - py * WG_TW;
      const int gy = y0 + py, gx = x0 + pxx, c = cob + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < p.H && gx < p.W) {
        const float* src = p.g + (((size_t)nb * p.H + gy) * p.W + gx) * p.Cg + c;
        if (c + 3 < p.Cg) v = *reinterpret_cast<const f32x4*>(src);                          // (Cg % 4 == 0 or a ragged tail)
        else { if (c < p.Cg) v.x = src[0]; if (c + 1 < p.Cg) v.y = src[1]; if (c + 2 < p.Cg) v.z = src[2]; }
      }
      rg[k] = v;
    }
  }
  __device__ __forceinline__ void store(const WgradArgs& p, float* sX, float* sG, int tid) {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int i = tid + 256 * k;
      if (i < (WG_TH + 2) * (WG_TW + 2) * 8) {
        f32x4 v = rx[k];
        if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(sX + (i >> 3) * 32 + 4 * (i & 7)) = v;
      }
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int i = tid + 256 * k;
      *reinterpret_cast<f32x4*>(sG + (i >> 3) * 32 + 4 * (i & 7)) = rg[k];
    }
  }
};

template <int LO, int HI> __device__ __forceinline__ void wg_put(float* dst, const f32x16 (&acc)[9], int lane) {
#pragma unroll
  for (int tap = LO; tap < HI; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((tap - LO) * 16 + r) * 64 + lane] = acc[tap][r];
}
template <int LO, int HI> __device__ __forceinline__ void wg_take(const float* src, f32x16 (&acc)[9], int lane) {
#pragma unroll
  for (int tap = LO; tap < HI; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tap][r] += src[((tap - LO) * 16 + r) * 64 + lane];
}

template <int WG_TW, int WG_TH>
__global__ __launch_bounds__(256, 2) void train_wgrad_kernel(const WgradArgs p) {
  static_assert(WG_TH % 4 == 0 && WG_TW % 2 == 0 && WG_TW * WG_TH <= 128 && WG_TW * WG_TH * 8 % 256 == 0, "tile geometry");
  static_assert(((WG_TH + 2) * (WG_TW + 2) + WG_TH * WG_TW) * 32 * 4 <= (int)WG_RED_BYTES, "tile LDS");
  extern __shared__ __attribute__((aligned(16))) char wg_smem[];
  float* const sX = reinterpret_cast<float*>(wg_smem);                       // [(TH+2) x (TW+2) px][32 ci]
  float* const sG = sX + (WG_TH + 2) * (WG_TW + 2) * 32;                     // [TH x TW px][32 co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int nci = (p.ci + 31) / 32, nco = (p.co + 31) / 32;
  const int vid = wg_xcd_order(blockIdx.x, gridDim.x);
  const int blk = vid % (nci * nco), ks = vid / (nci * nco);
  const int cib = (blk / nco) * 32, cob = (blk % nco) * 32;
  const int tiles_x = (p.W + WG_TW - 1) / WG_TW, tiles_y = (p.H + WG_TH - 1) / WG_TH;
  const int ntiles = tiles_x * tiles_y * p.N;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;                                   // bias gradient: every g value passes through a lane as operand B
  // Software pipeline: the global loads of the NEXT tile are issued into registers before the MFMAs of this one and
  // written to LDS after them, so their latency (eleven + eight dependent round trips otherwise) hides under the MFMAs.
  WgStager<WG_TW, WG_TH> stg;
  auto load_tile = [&](int tile) { stg.load(p, tile, tiles_x, tiles_y, cib, cob, tid); };
  auto store_tile = [&]() { stg.store(p, sX, sG, tid); };
  unsigned long long t_start = 0, t_mfma = 0, t_stage = 0, t_mark = 0, t_real = 0;
  if (p.trace) { t_start = t_mark = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }
  if (ks < ntiles) load_tile(ks);
  for (int tile = ks; tile < ntiles; tile += p.ksplit) {
    if (!((FISR_GABL & 1) && tile != ks)) {
    __syncthreads();                                 // every wave is done with the previous tile
    store_tile();
    __syncthreads();
    if (tile + p.ksplit < ntiles) load_tile(tile + p.ksplit);
    }
    if (FISR_GABL & 2) continue;
    if (p.trace) { const unsigned long long t = __builtin_readcyclecounter(); t_stage += t - t_mark; t_mark = t; }
    // wave w: tile rows w * TH/4 ..; K steps of two neighbouring pixels.  The ten LDS operands of step k+1 are requested
    // before the nine MFMAs of step k (the compiler's own schedule waited for each read right in front of its MFMA: the
    // matrix pipe was busy 64 % of the time on the largest layers).
    constexpr int NSTEP = (WG_TH / 4) * (WG_TW / 2);
    const float* const gx = sG + (wave * (WG_TH / 4) * WG_TW + kh) * 32 + li;
    const float* const ax = sX + (wave * (WG_TH / 4) * (WG_TW + 2) + kh) * 32 + li;
    auto fetch = [&](int step, float (&a)[9], float& b) {
      const int r = step / (WG_TW / 2), col = 2 * (step % (WG_TW / 2));
      b = gx[(r * WG_TW + col) * 32];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) a[tap] = ax[((r + tap / 3) * (WG_TW + 2) + col + tap % 3) * 32];
    };
    // (fully unrolled: every LDS offset is an immediate; the scheduling barriers keep the reads in front of the MFMAs)
    float av[2][9], bv[2];
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step + 1 < NSTEP) fetch(step + 1, av[(step + 1) & 1], bv[(step + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      bsum += bv[step & 1];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[step & 1][tap], bv[step & 1], acc[tap], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (p.trace) { const unsigned long long t = __builtin_readcyclecounter(); t_mfma += t - t_mark; t_mark = t; }
  }
  const unsigned long long t_loop = p.trace ? __builtin_readcyclecounter() : 0;
  // D layout: column (co) = lane & 31, row (ci) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int n = cob + li;
  // bias gradient: ONE atomic per output channel and workgroup (all 256 lanes sending theirs put 8 x ksplit x ... same-address
  // atomics on each of a few dozen words: 60 us of serialisation per launch on the 64-channel layers)
  bsum += __shfl_xor(bsum, 32);
  float* const sB = reinterpret_cast<float*>(wg_smem + WG_RED_BYTES);
  if ((FISR_GABL & 4) && acc[0][0] != 123.456f) return;
  // The four waves hold partial sums over different pixel rows.  Two exchange rounds through LDS (the tiles are dead by now; a
  // tap of one wave is 4 KB) leave every wave with the complete sums of two or three taps, and all four waves send their share
  // of the workgroup's 9216 atomics (one wave sending all of them spent 11 % of the workgroup's life doing so).
  //   round 1: waves 0 <-> 1 and 2 <-> 3 swap halves (the even wave keeps taps 0-4, the odd one taps 5-8)
  //   round 2: waves 0 <-> 2 (taps 0-2 / 3-4) and 1 <-> 3 (taps 5-6 / 7-8)
  float* const red = reinterpret_cast<float*>(wg_smem);
  constexpr int TAPF = 16 * 64;                      // floats per tap
  __syncthreads();
  if (wave == 0) wg_put<5, 9>(red, acc, lane);                    // 4 taps at 0
  if (wave == 1) wg_put<0, 5>(red + 4 * TAPF, acc, lane);         // 5 taps
  if (wave == 2) wg_put<5, 9>(red + 9 * TAPF, acc, lane);         // 4 taps
  if (wave == 3) wg_put<0, 5>(red + 13 * TAPF, acc, lane);        // 5 taps (18 taps = 72 KB in all)
  if (kh == 0) sB[wave * 32 + li] = bsum;
  __syncthreads();
  if (wave == 0) wg_take<0, 5>(red + 4 * TAPF, acc, lane);
  if (wave == 1) wg_take<5, 9>(red, acc, lane);
  if (wave == 2) wg_take<0, 5>(red + 13 * TAPF, acc, lane);
  if (wave == 3) wg_take<5, 9>(red + 9 * TAPF, acc, lane);
  if (wave == 0 && kh == 0 && p.db != nullptr && cib == 0 && n < p.co)
    unsafeAtomicAdd(p.db + n, (sB[li] + sB[32 + li]) + (sB[64 + li] + sB[96 + li]));
  __syncthreads();
  if (wave == 0) wg_put<3, 5>(red, acc, lane);                    // 2 taps at 0
  if (wave == 2) wg_put<0, 3>(red + 2 * TAPF, acc, lane);         // 3 taps
  if (wave == 1) wg_put<7, 9>(red + 5 * TAPF, acc, lane);         // 2 taps
  if (wave == 3) wg_put<5, 7>(red + 7 * TAPF, acc, lane);         // 2 taps
  __syncthreads();
  if (wave == 0) wg_take<0, 3>(red + 2 * TAPF, acc, lane);
  if (wave == 2) wg_take<3, 5>(red, acc, lane);
  if (wave == 1) wg_take<5, 7>(red + 7 * TAPF, acc, lane);
  if (wave == 3) wg_take<7, 9>(red + 5 * TAPF, acc, lane);
  const unsigned long long t_red = p.trace ? __builtin_readcyclecounter() : 0;
  const int tap_lo = wave == 0 ? 0 : (wave == 2 ? 3 : (wave == 1 ? 5 : 7)), tap_hi = wave == 0 ? 3 : (wave == 2 ? 5 : (wave == 1 ? 7 : 9));
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
    if (tap >= tap_lo && tap < tap_hi) {             // (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = cib + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (c < p.ci && n < p.co) unsafeAtomicAdd(p.dw + ((size_t)tap * p.ci + c) * p.co + n, acc[tap][r]);
      }
    }
  if (p.trace && tid == 0) {          // wave 0's view: start, staging / MFMA cycles of the tile loop, reduction, atomics issued
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_stage; tr[2] = t_mfma; tr[3] = t_loop; tr[4] = t_red; tr[5] = __builtin_readcyclecounter();
    tr[6] = t_real; tr[7] = __builtin_amdgcn_s_memrealtime();
  }
}

// The eight tile pairs of one wave and tile: operands transformed in registers, 8 MFMAs per pair (see train_wgrad_wino_kernel).
template <int PH, int TW, int NPAIR>
__device__ __forceinline__ void wg_wino_tile(const float* xb, const float* gb, f32x16 (&acc)[8], float& bsum) {
  constexpr int PPR = TW / 4;                        // tile pairs per Winograd-tile row (4 pixels per pair)
  // (Requesting the raw operands of pair q + 1 before the MFMAs of pair q -- fully unrolled, scheduling barriers -- was measured
  // and is SLOWER: 256 registers and spills, 148 -> 194 us on the 64 -> 64 layer at 96 x 96.  The compiler's own schedule stays.)
#if defined(FISR_WW_UNROLL) && FISR_WW_UNROLL == 1
#pragma unroll 1
#elif defined(FISR_WW_UNROLL) && FISR_WW_UNROLL == 4
#pragma unroll 4
#else
#pragma unroll 2
#endif
  for (int q = 0; q < NPAIR; ++q) {
    const float* xq = xb + ((q / PPR) * 2 * (TW + 2) + 4 * (q % PPR)) * 32;     // pair q: Winograd-tile row q / PPR, 4 pixels per pair
    const float* gq = gb + ((q / PPR) * 2 * TW + 4 * (q % PPR)) * 32;
    float d[3][4], y[2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = xq[(i * (TW + 2) + j) * 32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) y[i][j] = gq[(i * TW + j) * 32];
    // rows 2 PH, 2 PH + 1 of B^T d:  PH = 0: d0 - d2, d1 + d2 (patch rows 0, 1, 2);  PH = 1: d2 - d1, d1 - d3 (rows 1, 2, 3 = d[0..2])
    float u[2][4], e[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u[0][j] = PH == 0 ? d[0][j] - d[2][j] : d[1][j] - d[0][j];
      u[1][j] = PH == 0 ? d[1][j] + d[2][j] : d[0][j] - d[2][j];
    }
    // rows 2 PH, 2 PH + 1 of A dY:  PH = 0: y0, y0 + y1;  PH = 1: y0 - y1, -y1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      e[0][j] = PH == 0 ? y[0][j] : y[0][j] - y[1][j];
      e[1][j] = PH == 0 ? y[0][j] + y[1][j] : -y[1][j];
    }
    if (PH == 0) bsum += (y[0][0] + y[0][1]) + (y[1][0] + y[1][1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // (u B): u0 - u2, u1 + u2, u2 - u1, u1 - u3;   (e A^T): e0, e0 + e1, e0 - e1, -e1
      const float v0 = u[i][0] - u[i][2], v1 = u[i][1] + u[i][2], v2 = u[i][2] - u[i][1], v3 = u[i][1] - u[i][3];
      const float d0 = e[i][0], d1 = e[i][0] + e[i][1], d2 = e[i][0] - e[i][1], d3 = -e[i][1];
      acc[4 * i + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, d0, acc[4 * i + 0], 0, 0, 0);
      acc[4 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, d1, acc[4 * i + 1], 0, 0, 0);
      acc[4 * i + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, d2, acc[4 * i + 2], 0, 0, 0);
      acc[4 * i + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, d3, acc[4 * i + 3], 0, 0, 0);
    }
  }
}

// ---- weight gradient in the Winograd domain: F(3x3 weights <- 2x2 output-gradient tiles) ----
// The trilinear form sum y'_i g_k d_(i+k) that F(2x2,3x3) decomposes for the forward pass (Y = A^T[(G g G^T) . (B^T d B)]A) gives
// the weight gradient by the same rank-16 decomposition read the other way:
//     dW = G^T [ sum over tiles (A dY A^T) . (B^T X B) ] G
// -- 16 products per 2 x 2 output pixels instead of 36, the transforms of both operands are adds.  Per transform position p the
// sum over tiles is a GEMM M_p[ci][co] = sum_t V_p[ci][t] D_p[co][t] whose K is the TILE axis: v_mfma_f32_32x32x2_f32 with
// A = 32 input channels x 2 tiles, B = 2 tiles x 32 output channels.  The lane that owns (channel, tile) of an operand is the
// lane that computes its transform from the raw tiles in LDS, so V and D never exist in memory: 12 + 4 ds_read_b32 and ~30
// adds feed 8 MFMAs.  Workgroup = 4 rows x 32 pixels of one image = 2 x 16 Winograd tiles; wave (ph, r): transform rows
// 2 ph, 2 ph + 1 (8 of the 16 positions = 8 accumulators) of the tile row r.  After the tile loop: M G per wave (registers), the two
// tile rows summed through LDS, G^T across the two position halves through LDS, one atomic per weight and workgroup.
// Staging, work split and grid exactly as train_wgrad_kernel<TW, TH>; tiles of 4 x 32, 8 x 16, 16 x 8 or 8 x 8 pixels (the launcher
// takes the one that wastes the fewest pixels of the map): 32 (16) Winograd tiles, the wave's eight (four) pairs lie in 1, 2 or 4
// Winograd-tile rows.
template <int TW, int TH>
__global__ __launch_bounds__(256, 2) void train_wgrad_wino_kernel(const WgradArgs p) {
  static_assert((TW == 32 && TH == 4) || (TW == 16 && TH == 8) || (TW == 8 && TH == 16) || (TW == 8 && TH == 8), "tile geometry");
  constexpr int NPAIR = TW * TH / 16;                // tile pairs per wave and tile
  extern __shared__ __attribute__((aligned(16))) char wg_smem[];
  float* const sX = reinterpret_cast<float*>(wg_smem);                       // [(TH+2) x (TW+2) px][32 ci]
  float* const sG = sX + (TH + 2) * (TW + 2) * 32;                           // [TH x TW px][32 co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
  const int ph = wave & 1, r = wave >> 1;
  const int nci = (p.ci + 31) / 32, nco = (p.co + 31) / 32;
  const int vid = wg_xcd_order(blockIdx.x, gridDim.x);
  const int blk = vid % (nci * nco), ks = vid / (nci * nco);
  const int cib = (blk / nco) * 32, cob = (blk % nco) * 32;
  const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
  const int ntiles = tiles_x * tiles_y * p.N;
  f32x16 acc[8];                                      // position (2 ph + i, j) at acc[4 i + j]
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  float bsum = 0.f;
  WgStager<TW, TH> stg;
  unsigned long long t_start = 0, t_mfma = 0, t_stage = 0, t_mark = 0, t_real = 0;
  if (p.trace) { t_start = t_mark = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }
  if (ks < ntiles) stg.load(p, ks, tiles_x, tiles_y, cib, cob, tid);
  // this lane's Winograd tile of pair q (TW = 32): output pixels (2 r .. 2 r + 1, 2 (2 q + kh) ..), input patch rows 2 r .. 2 r + 3 of the halo tile
  // (wave r owns the Winograd-tile rows r TH/4 .. of the tile: output rows r TH/2 ..)
  const float* const xb = sX + ((r * (TH / 2) + ph) * (TW + 2) + 2 * kh) * 32 + li;   // first patch row this wave reads (ph: rows 0-2 / 1-3)
  const float* const gb = sG + (r * (TH / 2) * TW + 2 * kh) * 32 + li;
  for (int tile = ks; tile < ntiles; tile += p.ksplit) {
    __syncthreads();                                 // every wave is done with the previous tile
    stg.store(p, sX, sG, tid);
    __syncthreads();
    if (tile + p.ksplit < ntiles) stg.load(p, tile + p.ksplit, tiles_x, tiles_y, cib, cob, tid);
    if (p.trace) { const unsigned long long t = __builtin_readcyclecounter(); t_stage += t - t_mark; t_mark = t; }
    if (ph == 0) wg_wino_tile<0, TW, NPAIR>(xb, gb, acc, bsum);  // (wave-uniform)
    else wg_wino_tile<1, TW, NPAIR>(xb, gb, acc, bsum);
    if (p.trace) { const unsigned long long t = __builtin_readcyclecounter(); t_mfma += t - t_mark; t_mark = t; }
  }
  const unsigned long long t_loop = p.trace ? __builtin_readcyclecounter() : 0;
  // ---- M G (per wave): R[i][0] = M0 + (M1 + M2)/2, R[i][1] = (M1 - M2)/2, R[i][2] = (M1 + M2)/2 + M3 ----
  f32x16 R[6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float hs = 0.5f * (acc[4 * i + 1][q] + acc[4 * i + 2][q]), hd = 0.5f * (acc[4 * i + 1][q] - acc[4 * i + 2][q]);
      R[3 * i + 0][q] = acc[4 * i][q] + hs; R[3 * i + 1][q] = hd; R[3 * i + 2][q] = hs + acc[4 * i + 3][q];
    }
  const int n = cob + li;
  bsum += __shfl_xor(bsum, 32);
  float* const red = reinterpret_cast<float*>(wg_smem);
  float* const sB = reinterpret_cast<float*>(wg_smem + WG_RED_BYTES);
  constexpr int TAPF = 16 * 64;
  __syncthreads();                                   // the tiles are dead
  if (r == 1) {                                      // the second tile row's sums: 6 x 4 KB per wave, at ph * 24 KB
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) red[((ph * 6 + t) * 16 + q) * 64 + lane] = R[t][q];
  }
  if (kh == 0) sB[wave * 32 + li] = ph == 0 ? bsum : 0.f;
  __syncthreads();
  if (r == 0) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) R[t][q] += red[((ph * 6 + t) * 16 + q) * 64 + lane];
  }
  if (wave == 0 && kh == 0 && p.db != nullptr && cib == 0 && n < p.co)
    unsafeAtomicAdd(p.db + n, (sB[li] + sB[32 + li]) + (sB[64 + li] + sB[96 + li]));
  __syncthreads();
  // ---- G^T across the position halves: dW0 = R0 + R1/2 + R2/2, dW1 = R1/2 - R2/2, dW2 = R1/2 + R2/2 + R3 (rows; R0, R1 in wave
  // (0,0), R2, R3 in wave (1,0)): wave (1,0) hands over P = R2/2 and Q = R2/2 + R3 ----
  if (wave == 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float half = 0.5f * R[j][q];
        red[(j * 16 + q) * 64 + lane] = half;
        red[((3 + j) * 16 + q) * 64 + lane] = half + R[3 + j][q];
      }
  }
  __syncthreads();
  if (wave != 0) return;
  const unsigned long long t_red = p.trace ? __builtin_readcyclecounter() : 0;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float h1 = 0.5f * R[3 + j][q], P = red[(j * 16 + q) * 64 + lane], Q = red[((3 + j) * 16 + q) * 64 + lane];
      const float w0 = R[j][q] + h1 + P, w1 = h1 - P, w2 = h1 + Q;
      const int c = cib + (q & 3) + 8 * (q >> 2) + 4 * kh;
      if (c < p.ci && n < p.co) {
        unsafeAtomicAdd(p.dw + ((size_t)(0 + j) * p.ci + c) * p.co + n, w0);
        unsafeAtomicAdd(p.dw + ((size_t)(3 + j) * p.ci + c) * p.co + n, w1);
        unsafeAtomicAdd(p.dw + ((size_t)(6 + j) * p.ci + c) * p.co + n, w2);
      }
    }
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_stage; tr[2] = t_mfma; tr[3] = t_loop; tr[4] = t_red; tr[5] = __builtin_readcyclecounter();
    tr[6] = t_real; tr[7] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- weight gradient of the 3- and 6-channel heads (co <= 8, g padded to 16 channels) on the vector ALU ----
// On train_wgrad_kernel a head fills 3 or 6 of the 32 MFMA columns (64 -> 6 at 192 x 192 x 32 images: 408 us).  Here a lane
// owns one input channel and keeps all 9 x CO sums in registers; a wave walks along one image row with the 3 x 3 window of
// its channel in registers (eight columns at a time, the next eight already requested), the CO gradients of the pixel come
// from one 64-byte load and v_readlane as scalar FMA operands.  Four waves of a workgroup take different rows; their sums
// meet in LDS and leave as one atomic per weight and workgroup.
template <int CO>
__global__ __launch_bounds__(256) void train_wgrad_head_kernel(const WgradArgs p) {
  __shared__ float s_red[9 * CO * 64];
  __shared__ float s_b[4][CO];
  constexpr int CH = 8;                                        // columns per register chunk
  constexpr int HC0 = 64, HCG = 16;                            // pixel strides of x and g (what the heads have; the launcher checks)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nci = (p.ci + 63) / 64;
  const int blk = blockIdx.x % nci, ks = blockIdx.x / nci;
  const int c = blk * 64 + lane;
  const bool cok = c < HC0;
  float acc[9][CO];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[t][o] = 0.f;
  float bs = 0.f;
  const int units = p.N * p.H;
  for (int u = ks * 4 + wave; u < units; u += p.ksplit * 4) {
    const int n = u / p.H, y = u - n * p.H;
    const float* xr[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      xr[dy] = (cok && yy >= 0 && yy < p.H) ? p.x0 + ((size_t)(n * p.H + yy) * p.W) * HC0 + c : nullptr;
    }
    const float* gr = lane < HCG ? p.g + ((size_t)(n * p.H + y) * p.W) * HCG + lane : nullptr;
    float cur[3][CH], nxt[3][CH], gc[CH], gn[CH], left[3] = {0.f, 0.f, 0.f};
    // (the pixel strides are compile-time constants, HC0 and HCG: every load of a chunk is base + immediate offset)
    auto fetch = [&](int x0, float (&xv)[3][CH], float (&gv)[CH]) {
      const float* xb[3] = {xr[0] ? xr[0] + (size_t)x0 * HC0 : nullptr, xr[1] ? xr[1] + (size_t)x0 * HC0 : nullptr,
                            xr[2] ? xr[2] + (size_t)x0 * HC0 : nullptr};
      const float* gb = gr ? gr + (size_t)x0 * HCG : nullptr;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const bool in = x0 + j < p.W;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) xv[dy][j] = (in && xb[dy]) ? xb[dy][j * HC0] : 0.f;
        gv[j] = (in && gb) ? gb[j * HCG] : 0.f;
      }
    };
    fetch(0, cur, gc);
    for (int x0 = 0; x0 < p.W; x0 += CH) {
      fetch(x0 + CH, nxt, gn);                                 // (all zeros behind the row's end)
      if (p.relu_in) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int j = 0; j < CH; ++j) cur[dy][j] = fmaxf(cur[dy][j], 0.f);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (lane < CO) bs += gc[j];
        float go[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) go[o] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gc[j]), o));
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float l = j == 0 ? left[dy] : cur[dy][j - 1];
          const float r = j == CH - 1 ? (p.relu_in ? fmaxf(nxt[dy][0], 0.f) : nxt[dy][0]) : cur[dy][j + 1];
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            acc[dy * 3 + 0][o] = fmaf(l, go[o], acc[dy * 3 + 0][o]);
            acc[dy * 3 + 1][o] = fmaf(cur[dy][j], go[o], acc[dy * 3 + 1][o]);
            acc[dy * 3 + 2][o] = fmaf(r, go[o], acc[dy * 3 + 2][o]);
          }
        }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        left[dy] = cur[dy][CH - 1];
#pragma unroll
        for (int j = 0; j < CH; ++j) cur[dy][j] = nxt[dy][j];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) gc[j] = gn[j];
    }
  }
  // the four waves' sums -> LDS -> one atomic per weight
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int o = 0; o < CO; ++o) s_red[(t * CO + o) * 64 + lane] = acc[t][o];
  }
  if (lane < CO) s_b[wave][lane] = bs;
  __syncthreads();
  if (wave != 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int o = 0; o < CO; ++o) atomicAdd(&s_red[(t * CO + o) * 64 + lane], acc[t][o]);
  }
  __syncthreads();
  for (int i = tid; i < 9 * CO * 64; i += 256) {
    const int l = i & 63, to = i >> 6, o = to % CO, t = to / CO;
    const int cc = blk * 64 + l;
    if (cc < p.ci && o < p.co) unsafeAtomicAdd(p.dw + ((size_t)t * p.ci + cc) * p.co + o, s_red[i]);
  }
  if (p.db != nullptr && blk == 0 && tid < CO && tid < p.co) unsafeAtomicAdd(p.db + tid, (s_b[0][tid] + s_b[1][tid]) + (s_b[2][tid] + s_b[3][tid]));
}

// db[c] += sum_p g[p][c]   (grid (pixel blocks, ceil(C / 64)); 256 threads = 64 channels x 4 pixel phases)
__global__ void train_bgrad_kernel(const float* __restrict__ g, int C, size_t npix, float* __restrict__ db, int co) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  if (c >= C) return;
  float s = 0.f;
  for (size_t pix = (size_t)blockIdx.x * 4 + sub; pix < npix; pix += (size_t)gridDim.x * 4) s += g[pix * C + c];
  if (c < co) unsafeAtomicAdd(db + c, s);
}

// g_out = g_in * (ref > 0)    (in place allowed)
__global__ void train_relu_bwd_kernel(const float* __restrict__ g_in, const float* __restrict__ ref, float* __restrict__ g_out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 g = reinterpret_cast<const f32x4*>(g_in)[i], r = reinterpret_cast<const f32x4*>(ref)[i];
    f32x4 o;
    o.x = r.x > 0.f ? g.x : 0.f; o.y = r.y > 0.f ? g.y : 0.f; o.z = r.z > 0.f ? g.z : 0.f; o.w = r.w > 0.f ? g.w : 0.f;
    reinterpret_cast<f32x4*>(g_out)[i] = o;
  }
}

// y += a * x
__global__ void train_axpy_kernel(const float* __restrict__ x, float a, float* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o = reinterpret_cast<f32x4*>(y)[i];
    o.x += a * v.x; o.y += a * v.y; o.z += a * v.z; o.w += a * v.w;
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
}

// 2x2 max-pool backward (ops.py:54): the gradient of a window goes to its first maximum in row-major order.
// x [N,H,W,C], dpool [N,H/2,W/2,C] -> dx [N,H,W,C] (every element written)
__global__ void train_maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dp, float* __restrict__ dx,
                                          int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, C4 = C / 4;
  const size_t total = (size_t)N * OH * OW * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t t = i / C4;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    const size_t base = (((size_t)n * H + 2 * oy) * W + 2 * ox) * C + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + base), b = *reinterpret_cast<const f32x4*>(x + base + C);
    const f32x4 d = *reinterpret_cast<const f32x4*>(x + base + (size_t)W * C), e = *reinterpret_cast<const f32x4*>(x + base + (size_t)W * C + C);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dp + (((size_t)n * OH + oy) * OW + ox) * C + c);
    f32x4 ga, gb, gd, ge;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float m = fmaxf(fmaxf(a[k], b[k]), fmaxf(d[k], e[k]));
      const int w = a[k] == m ? 0 : (b[k] == m ? 1 : (d[k] == m ? 2 : 3));
      ga[k] = w == 0 ? g[k] : 0.f; gb[k] = w == 1 ? g[k] : 0.f; gd[k] = w == 2 ? g[k] : 0.f; ge[k] = w == 3 ? g[k] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dx + base) = ga; *reinterpret_cast<f32x4*>(dx + base + C) = gb;
    *reinterpret_cast<f32x4*>(dx + base + (size_t)W * C) = gd; *reinterpret_cast<f32x4*>(dx + base + (size_t)W * C + C) = ge;
  }
}

// Adjoint of the legacy x2 bilinear (ops.py:69; glue_kernels.h upsample2): out[2i] = x[i], out[2i+1] = x[i] + (x[i'] - x[i]) / 2,
// i' = min(i + 1, last), along W first and then along H.  Hence per axis dx[i] = d[2i] + d[2i+1] / 2 + d[2i-1] / 2 (i >= 1)
// and the last sample also keeps the second half of its own odd neighbour.  dy [N,2H,2W,C] -> dx [N,H,W,C].
__global__ void train_upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    // weights of the (up to) 3 contributing fine samples per axis: fine index 2i-1, 2i, 2i+1
    float wy[3] = {y > 0 ? 0.5f : 0.f, 1.f, y == H - 1 ? 1.f : 0.5f};
    float wx[3] = {x > 0 ? 0.5f : 0.f, 1.f, x == W - 1 ? 1.f : 0.5f};
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (wy[a] == 0.f) continue;
      const int fy = 2 * y - 1 + a;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        if (wx[b] == 0.f) continue;
        const int fx = 2 * x - 1 + b;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (((size_t)n * 2 * H + fy) * (2 * W) + fx) * C + c);
        const float w = wy[a] * wx[b];
        s.x += w * v.x; s.y += w * v.y; s.z += w * v.z; s.w += w * v.w;
      }
    }
    *reinterpret_cast<f32x4*>(dx + (((size_t)n * H + y) * W + x) * C + c) = s;
  }
}

// adjoint of tf.depth_to_space(x, 2) (FISRnet.py:99): out[h, w, (2i + j) * C + c] = g[2h + i, 2w + j, c]
__global__ void train_s2d_kernel(const float* __restrict__ g, float* __restrict__ out, int N, int H, int W, int C) {
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * 4 * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t t = i / C4;
    const int sub = (int)(t & 3); t >>= 2;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + (((size_t)n * 2 * H + 2 * y + (sub >> 1)) * (2 * W) + 2 * x + (sub & 1)) * C + c);
    *reinterpret_cast<f32x4*>(out + (((size_t)n * H + y) * W + x) * 4 * C + sub * C + c) = v;
  }
}

// dst[p][dco + k] (op)= src[p][sco + k], k < nc   (channel ranges of NHWC tensors; add != 0 accumulates)
__global__ void train_copy_channels_kernel(const float* __restrict__ src, int scs, int sco, float* __restrict__ dst, int dcs, int dco,
                                           int nc, size_t npix, int add) {
  const size_t total = npix * nc;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % nc);
    const size_t pix = i / nc;
    const float v = src[pix * scs + sco + k];
    float* d = dst + pix * dcs + dco + k;
    *d = add ? *d + v : v;
  }
}

// ---- the seven loss terms of one level and their gradients (FISRnet.py:316-484) ----
// pred[k], k = 0..2: the three stride-1 windows' outputs [B,h,w,9]; pred[3]: the stride-2 window's; gt [B,h,w,21].
// Per element (pixel, colour c): p[3k + f] = pred[k][3f + c] (nine stride-1 frames), q[f] = pred[3][3f + c], t[f] = gt[3f + c].
// L2_loss = mean over ITS tensor: n3 = B*h*w*9 for three-frame terms, n1 = B*h*w*3 for one-frame terms.
// sums[0..6] += raw squared sums of recn, tm, tmm, td, recn_ss2, td_ss2, tm_ss2 (host applies the 1/n, level scale, lambdas).
struct LossArgs {
  const float* pred[4]; const float* gt; float* grad[4]; float* sums; size_t npix;
  float k_recn, k_tm, k_tmm, k_td, k_recn2, k_td2, k_tm2;      // lambda * level scale * 2 / n of each term
};
__global__ void train_loss_kernel(const LossArgs a) {
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t total = a.npix * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const size_t pix = i / 3;
    float p[9], q[3], t[7], gp[9], gq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int f = 0; f < 3; ++f) p[3 * k + f] = a.pred[k][pix * 9 + 3 * f + c];
#pragma unroll
    for (int f = 0; f < 3; ++f) q[f] = a.pred[3][pix * 9 + 3 * f + c];
#pragma unroll
    for (int f = 0; f < 7; ++f) t[f] = a.gt[pix * 21 + 3 * f + c];
#pragma unroll
    for (int k = 0; k < 9; ++k) gp[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) gq[k] = 0.f;
    // type 1: window k, frame f against GT frame 2k + f
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int f = 0; f < 3; ++f) { const float d = p[3 * k + f] - t[2 * k + f]; acc[0] += d * d; gp[3 * k + f] += a.k_recn * d; }
    // types 2, 3: the overlapped frames p[3k+2], p[3k+3] and GT frame 2(k+1)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float d = p[3 * k + 2] - p[3 * k + 3];
      acc[1] += d * d; gp[3 * k + 2] += a.k_tm * d; gp[3 * k + 3] -= a.k_tm * d;
      const float m = (p[3 * k + 2] + p[3 * k + 3]) * 0.5f - t[2 * (k + 1)];
      acc[2] += m * m; gp[3 * k + 2] += a.k_tmm * 0.5f * m; gp[3 * k + 3] += a.k_tmm * 0.5f * m;
    }
    // Groups2Ovlp (ops.py:119-144) and its adjoint
    float o[7] = {p[0], p[1], (p[2] + p[3]) * 0.5f, p[4], (p[5] + p[6]) * 0.5f, p[7], p[8]}, go[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // type 4: temporal differences of the overlapped sequence
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float d = (o[k + 1] - o[k]) - (t[k + 1] - t[k]); acc[3] += d * d; go[k + 1] += a.k_td * d; go[k] -= a.k_td * d; }
    // stride 2: q against GT frames 1, 3, 5 (type 5), its differences (type 6), against the overlapped frames 1, 3, 5 (type 7)
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const float d = q[f] - t[2 * f + 1]; acc[4] += d * d; gq[f] += a.k_recn2 * d;
      const float e = q[f] - o[2 * f + 1]; acc[6] += e * e; gq[f] += a.k_tm2 * e; go[2 * f + 1] -= a.k_tm2 * e;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float d = (q[f + 1] - q[f]) - (t[2 * f + 3] - t[2 * f + 1]); acc[5] += d * d; gq[f + 1] += a.k_td2 * d; gq[f] -= a.k_td2 * d;
    }
    gp[0] += go[0]; gp[1] += go[1]; gp[2] += 0.5f * go[2]; gp[3] += 0.5f * go[2]; gp[4] += go[3];
    gp[5] += 0.5f * go[4]; gp[6] += 0.5f * go[4]; gp[7] += go[5]; gp[8] += go[6];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int f = 0; f < 3; ++f) a.grad[k][pix * 9 + 3 * f + c] = gp[3 * k + f];
#pragma unroll
    for (int f = 0; f < 3; ++f) a.grad[3][pix * 9 + 3 * f + c] = gq[f];
  }
  // one atomic per term and WORKGROUP (the seven sums share a cache line: one atomic per wave of a 3456-workgroup grid
  // serialised into 1.2 ms on the largest level)
  __shared__ float s_part[4][8];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    float v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7) unsafeAtomicAdd(a.sums + threadIdx.x, (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]));
}

// tf.train.AdamOptimizer (TF 1.13 training/adam.py): m, v, var updated in place; lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) from the host
__global__ void train_adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                  size_t n, float lr_t, float b1, float b2, float eps) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    w[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

}  // namespace fisr
