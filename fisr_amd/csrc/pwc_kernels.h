// On-GPU optical flow for `--phase FISR_for_video` (cfg5 of BASELINE.json): the kernels of PWC-Net-large as the
// reference runs it (FISR_tfoptflow/model_pwcnet.py:1525-1593, options of
// FISR_for_video_pwcnet_predict_from_img_test.py:96-100) plus the script's own pre/post-processing (:118-140).
// All fp32, NHWC.  Channel groups inside the concatenated feature buffers are padded to multiples of 4 channels
// (zero weights on the padding), so every access is a 16-byte vector.
//
//   pwc_convg_kernel      tf.layers.conv2d 3x3 'same', stride 1|2, dilation d, + bias (+ add) + leaky relu, reading a
//                         channel RANGE of a wider NHWC buffer and writing into a channel range of another: the dense
//                         blocks' tf.concat([act, x]) (model_pwcnet.py:1428-1445) never materialise.  Implicit GEMM on
//                         v_mfma_f32_32x32x2_f32, 8x32 output pixels x 64 channels per workgroup, K chunks of 8 input
//                         channels with all 9 tap-shifted pixel blocks staged in LDS (stride / dilation / padding are
//                         resolved by the loader, the MFMA loop is the same for every layer).
//   pwc_deconv_kernel     tf.layers.conv2d_transpose(x, 2, 4, 2, 'same') (:1196)
//   pwc_costvol_kernel    core_costvol.cost_volume: 81 displacements, mean over channels, leaky relu (:1277); w2 tiled
//                         through LDS
//   pwc_warp_kernel       core_warp.dense_image_warp: bilinear sample at (x + u, y + v), clamped (:1178)
//   pwc_prep_kernel       script :121-131 + adapt_x (:399-411): YUV uint8 -> RGB (double), x2 up-resize as
//                         scikit-image does it, uint8 truncation, / 255, zero pad to a multiple of 64
//   pwc_flow_out_kernel   :1587-1590 + script :139: x4 legacy bilinear * 4, crop, anti-aliased /2 down-resize as
//                         scikit-image does it (Gaussian sigma 0.5, 'mirror'), / 2
#pragma once
#include "glue_kernels.h"

namespace fisr {

// Element type of the flow network's feature tensors: float (the exact engine) or _Float16 (FISR_PREC_F16: 16-bit features,
// fp32 arithmetic inside these kernels, fp32 flows).  Four consecutive channels <-> one f32x4.
template <typename T> struct PwcElem;
template <> struct PwcElem<float> {
  static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct PwcElem<_Float16> {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4 ld4(const _Float16* p) {
    const h4 h = *reinterpret_cast<const h4*>(p);
    return f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
  }
  static __device__ __forceinline__ void st4(_Float16* p, f32x4 v) {
    h4 h; h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
    *reinterpret_cast<h4*>(p) = h;
  }
};

// (pair, direction) items of a batched decoder pass: item i correlates the features of frame a[i] with those of frame b[i]
constexpr int PWC_MAX_ITEMS = 8;
struct PwcItems { int n; int a[PWC_MAX_ITEMS]; int b[PWC_MAX_ITEMS]; };

constexpr int G_CH = 8;                 // input channels per K chunk
constexpr int G_REC = 32;               // bytes per LDS record
constexpr int G_PX = 256;               // output pixels per workgroup (8 x 32)
constexpr int G_BN = 64;                // output channels per workgroup
constexpr size_t convg_lds_bytes() { return (size_t)9 * G_PX * G_REC + (size_t)9 * G_BN * G_REC; }

struct ConvGArgs {
  const void* in;   int in_cs, in_co, Cin;      // input buffer (TI): pixel stride (elements), first channel, channels read
  const float* w;                                // packed [Cin8/8][9][CoutPad][8], LDS image (halves swizzled by row bit 3)
  const float* bias;                             // [CoutPad]
  void* out;        int out_cs, out_co, Cout, CoutPad;     // (TO)
  const float* add; int add_cs, add_co;          // optional: out = act(conv + bias) + add   (refine_flow, :1521)
  int N, H, W, OH, OW, stride, dil, pad_t, pad_l;
  float slope;                                   // leaky relu slope (1 = linear)
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256, 1) void pwc_convg_kernel(const ConvGArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sIn = smem;                               // [tap 9][px 256][32 B]
  char* const sW = smem + 9 * G_PX * G_REC;             // [tap 9][row 64][32 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int pxh = wave & 1, coh = wave >> 1;            // wave = 128 pixels (4 rows) x 32 channels

  const int tiles_x = (p.OW + TILE_W - 1) / TILE_W, tiles_y = (p.OH + TILE_H - 1) / TILE_H;
  const int nblocks = p.CoutPad / G_BN;
  int t = blockIdx.x / nblocks;
  const int nblk = blockIdx.x - t * nblocks;
  const int tx_ = t % tiles_x; t /= tiles_x;
  const int ty_ = t % tiles_y;
  const int nb = t / tiles_y;
  const int x0 = tx_ * TILE_W, y0 = ty_ * TILE_H;
  const int nch = (p.Cin + G_CH - 1) / G_CH;

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // loader geometry: unit u = tid + 256*i, i < 18: half = u & 1, pixel = (u >> 1) & 255, tap = u >> 9
  const int l_half = tid & 1, l_px = tid >> 1;          // + 128*(i & 1) pixels, tap = i >> 1
  const int f_off = li * G_REC + ((kh ^ ((li >> 3) & 1)) * 16);

  for (int kc = 0; kc < nch; ++kc) {
    __syncthreads();                                    // previous chunk's fragments are consumed
    const int c0 = kc * G_CH + 4 * l_half;
    const bool c_ok = c0 < p.Cin;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int px = l_px + 128 * (i & 1), tap = i >> 1;
      const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
      const int iy = oy * p.stride - p.pad_t + (tap / 3) * p.dil, ix = ox * p.stride - p.pad_l + (tap % 3) * p.dil;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
        v = PwcElem<TI>::ld4((const TI*)p.in + ((size_t)(nb * p.H + iy) * p.W + ix) * p.in_cs + p.in_co + c0);
      *reinterpret_cast<f32x4*>(sIn + (tap * G_PX + px) * G_REC + ((l_half ^ ((px >> 3) & 1)) * 16)) = v;
    }
    {
      const char* g = (const char*)p.w + ((size_t)kc * nblocks + nblk) * (9 * G_BN * G_REC);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int u = tid + 256 * i;
        if (u < 9 * G_BN * 2) *reinterpret_cast<uint4*>(sW + u * 16) = *reinterpret_cast<const uint4*>(g + (size_t)u * 16);
      }
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(sW + (tap * G_BN + 32 * coh) * G_REC + f_off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(sIn + (tap * G_PX + 32 * (4 * pxh + j)) * G_REC + f_off);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
      }
    }
  }
  // epilogue: lane (li, kh) of block j owns pixel (y0 + 4*pxh + j, x0 + li), channels cb + r, r = 0..15
  const int cb = nblk * G_BN + 32 * coh + 16 * kh;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = p.bias[cb + r];
  const int ox = x0 + li;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = y0 + 4 * pxh + j;
    if (oy >= p.OH || ox >= p.OW || cb >= p.Cout) continue;
    const size_t pix = (size_t)(nb * p.OH + oy) * p.OW + ox;
    TO* ob = (TO*)p.out + pix * p.out_cs + p.out_co + cb;
    const float* ab = p.add ? p.add + pix * p.add_cs + p.add_co + cb : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = acc[j][4 * q + e] + bv[4 * q + e];
        v[e] = s >= 0.f ? s : s * p.slope;
      }
      if (cb + 4 * q + 3 < p.Cout) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        if (ab) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ab + 4 * q); o += a4; }
        PwcElem<TO>::st4(ob + 4 * q, o);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + 4 * q + e < p.Cout) ob[4 * q + e] = (TO)(v[e] + (ab ? ab[4 * q + e] : 0.f));
      }
    }
  }
}

// The same implicit GEMM in the fp16 engine's arithmetic (r05): fp16 features and fp16 weights on v_mfma_f32_32x32x16_f16, fp32
// accumulation.  pwc_convg_kernel<_Float16, _Float16> converted every loaded value to fp32 and multiplied on the fp32 pipe -- 144 MFMAs
// of 64 cycles per wave and 8-channel chunk, one workgroup per CU, loads and MFMAs one after the other: the stride-2 pyramid
// convolutions of a 5-frame stack took 2.45 ms for 0.8 GB of traffic.  Here a chunk is 16 channels in the same 32-byte records (9
// tap-shifted pixel blocks + the weight slab) and costs 18 MFMAs of 32 cycles per wave; a workgroup is 4 x 32 output pixels x 64
// channels with 54 KB of LDS, so two of them share a CU and one's loads fly under the other's MFMAs.
// A16: Cin % 16 == 0 and 16-byte aligned pixel records (in_cs % 8 == 0, in_co % 8 == 0) -> one 16-byte load per unit; otherwise
// (the 196-channel maps of pyramid level 6: in_cs % 4 == 0, in_co % 4 == 0, Cin % 4 == 0) two 8-byte loads, the second of the last chunk guarded.
// w: [ceil(Cin/16)][CoutPad/64][9][64 rows][16 halves], halves of a record swizzled by row bit 3, zeros behind Cin (pwc_pack_conv).
constexpr int G16_TH = 4, G16_PX = G16_TH * 32;       // output pixels of a workgroup
constexpr size_t convg16_lds_bytes() { return (size_t)9 * G16_PX * G_REC + (size_t)9 * G_BN * G_REC; }     // 55296
template <bool A16>
__global__ __launch_bounds__(256, 2) void pwc_convg_f16_kernel(const ConvGArgs p) {
  typedef _Float16 TE;
  constexpr int CH = 16, NU = 9 * G16_PX * 2 / 256;     // 9 loader units per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sIn = smem;                               // [tap 9][px 128][32 B]
  char* const sW = smem + 9 * G16_PX * G_REC;           // [tap 9][row 64][32 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int pxh = wave & 1, coh = wave >> 1;            // wave = 64 pixels (2 rows) x 32 channels

  const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + G16_TH - 1) / G16_TH;
  const int nblocks = p.CoutPad / G_BN;
  int t = blockIdx.x / nblocks;
  const int nblk = blockIdx.x - t * nblocks;
  const int tx_ = t % tiles_x; t /= tiles_x;
  const int ty_ = t % tiles_y;
  const int nb = t / tiles_y;
  const int x0 = tx_ * 32, y0 = ty_ * G16_TH;
  const int nch = (p.Cin + CH - 1) / CH;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // loader: unit u = tid + 256 i, i < 9: half = u & 1, pixel = (u >> 1) & 127, tap = u >> 8 = i.  The source offsets do not depend on
  // the chunk: computed once (pixel index in the image, or -1 outside it)
  const int l_half = tid & 1, l_px = tid >> 1;
  const int f_off = li * G_REC + ((kh ^ ((li >> 3) & 1)) * 16);
  int src[NU];
  {
    const int oy = y0 + (l_px >> 5), ox = x0 + (l_px & 31);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int iy = oy * p.stride - p.pad_t + (i / 3) * p.dil, ix = ox * p.stride - p.pad_l + (i % 3) * p.dil;
      src[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (iy * p.W + ix) : -1;
    }
  }
  const TE* const img = (const TE*)p.in + (size_t)nb * p.H * p.W * p.in_cs + p.in_co + 8 * l_half;
  char* const s_dst = sIn + l_px * G_REC + ((l_half ^ ((l_px >> 3) & 1)) * 16);
  const bool w5 = tid + 1024 < 9 * G_BN * 2;            // 1152 weight units: four per thread + one for the first 128 threads

  for (int kc = 0; kc < nch; ++kc) {
    uint4 v[NU], wv[5];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      v[i] = make_uint4(0u, 0u, 0u, 0u);
      if (src[i] >= 0) {
        const TE* q = img + (size_t)src[i] * p.in_cs + kc * CH;
        if constexpr (A16) v[i] = *reinterpret_cast<const uint4*>(q);
        else {
          const int c = kc * CH + 8 * l_half;                // first channel of this unit
          uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
          if (c < p.Cin) lo = *reinterpret_cast<const uint2*>(q);
          if (c + 4 < p.Cin) hi = *reinterpret_cast<const uint2*>(q + 4);
          v[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    }
    const uint4* g = reinterpret_cast<const uint4*>((const char*)p.w + ((size_t)kc * nblocks + nblk) * (9 * G_BN * G_REC)) + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = g[256 * i];
    wv[4] = w5 ? g[1024] : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();                                    // the previous chunk's fragments are consumed (the loads above are in flight)
#pragma unroll
    for (int i = 0; i < NU; ++i) *reinterpret_cast<uint4*>(s_dst + i * (G16_PX * G_REC)) = v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sW + (tid + 256 * i) * 16) = wv[i];
    if (w5) *reinterpret_cast<uint4*>(sW + (tid + 1024) * 16) = wv[4];
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(sW + (tap * G_BN + 32 * coh) * G_REC + f_off);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f16x8 b = *reinterpret_cast<const f16x8*>(sIn + (tap * G16_PX + 32 * (2 * pxh + j)) * G_REC + f_off);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
      }
    }
  }
  // epilogue: lane (li, kh) of block j owns pixel (y0 + 2*pxh + j, x0 + li), channels cb + r, r = 0..15
  const int cb = nblk * G_BN + 32 * coh + 16 * kh;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = p.bias[cb + r];
  const int ox = x0 + li;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int oy = y0 + 2 * pxh + j;
    if (oy >= p.OH || ox >= p.OW || cb >= p.Cout) continue;
    const size_t pix = (size_t)(nb * p.OH + oy) * p.OW + ox;
    TE* ob = (TE*)p.out + pix * p.out_cs + p.out_co + cb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s_ = acc[j][4 * q + e] + bv[4 * q + e];
        v4[e] = s_ >= 0.f ? s_ : s_ * p.slope;
      }
      if (cb + 4 * q + 3 < p.Cout) {
        PwcElem<TE>::st4(ob + 4 * q, f32x4{v4[0], v4[1], v4[2], v4[3]});
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cb + 4 * q + e < p.Cout) ob[4 * q + e] = (TE)v4[e];
      }
    }
  }
}

// tf.layers.conv2d_transpose(x, 2, 4, 2, 'same'): out[2*i + k - 1] += in[i] * kernel[k]  (two output channels).
// EIGHT lanes per output pixel, each taking every eighth group of 4 input channels (the eight lanes of a pixel read 32
// consecutive channels per step: whole 64- / 128-byte pieces of a pixel record instead of one lane striding through a 1.2-KB
// record alone), then a 3-step butterfly; weights [ky][kx][o][Cin4] in global (L1/L2 resident), 16-byte loads.
// (r03: one thread per pixel took 7.7 ms per launch on the 608-channel level-2 buffer of a 5-frame 1080p stack.)
template <typename TI, typename TO>
__global__ void pwc_deconv_kernel(const TI* __restrict__ in, int in_cs, int in_co, int Cin4, const float* __restrict__ w,
                                  const float* __restrict__ bias, TO* __restrict__ out, int out_cs, int out_co,
                                  int N, int H, int W, int pad4) {
  const int OH = 2 * H, OW = 2 * W;
  const size_t total = (size_t)N * OH * OW;
  const int sub = threadIdx.x & 7;
  const size_t gstride = ((size_t)gridDim.x * blockDim.x) >> 3;
  const size_t rounds = (total + gstride - 1) / gstride;          // every lane runs every round: the butterfly needs all eight
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  for (size_t r_ = 0; r_ < rounds; ++r_, i += gstride) {
    const bool live = i < total;
    const size_t ii = live ? i : total - 1;
    const int ox = (int)(ii % OW), oy = (int)((ii / OW) % OH), n = (int)(ii / ((size_t)OW * OH));
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const int ky = ((oy + 1) & 1) + 2 * ty, iy = (oy + 1 - ky) / 2;
      if ((oy + 1 - ky) < 0 || iy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int kx = ((ox + 1) & 1) + 2 * tx, ix = (ox + 1 - kx) / 2;
        if ((ox + 1 - kx) < 0 || ix >= W) continue;
        const TI* src = in + ((size_t)(n * H + iy) * W + ix) * in_cs + in_co;
        const float* w0 = w + ((size_t)(ky * 4 + kx) * 2 + 0) * Cin4;
        const float* w1 = w + ((size_t)(ky * 4 + kx) * 2 + 1) * Cin4;
        for (int c = 4 * sub; c < Cin4; c += 32) {
          const f32x4 v = PwcElem<TI>::ld4(src + c);
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(w0 + c), k1 = *reinterpret_cast<const f32x4*>(w1 + c);
          a0 += v.x * k0.x + v.y * k0.y + v.z * k0.z + v.w * k0.w;
          a1 += v.x * k1.x + v.y * k1.y + v.z * k1.z + v.w * k1.w;
        }
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); }
    if (live && sub == 0) {
      TO* o = out + i * out_cs + out_co;
      // pad4: the two padding channels behind the pair (the decoder buffers' groups are multiples of 4 channels) as part of one store
      if (pad4) PwcElem<TO>::st4(o, f32x4{a0 + bias[0], a1 + bias[1], 0.f, 0.f});
      else { o[0] = (TO)(a0 + bias[0]); o[1] = (TO)(a1 + bias[1]); }
    }
  }
}

// The same transpose conv for a FOUR-channel input (up_flow: the 2-channel flow in a 4-channel record): one thread per output pixel --
// the eight-lane form above leaves seven of its lanes without a channel group (325 us at level 2 of a 5-frame stack).  Same taps in the
// same order: bit-identical sums.
template <typename TI, typename TO>
__global__ void pwc_deconv4_kernel(const TI* __restrict__ in, int in_cs, int in_co, const float* __restrict__ w, const float* __restrict__ bias,
                                   TO* __restrict__ out, int out_cs, int out_co, int N, int H, int W, int pad4) {
  const int OH = 2 * H, OW = 2 * W;
  const size_t total = (size_t)N * OH * OW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), n = (int)(i / ((size_t)OW * OH));
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const int ky = ((oy + 1) & 1) + 2 * ty, iy = (oy + 1 - ky) / 2;
      if ((oy + 1 - ky) < 0 || iy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int kx = ((ox + 1) & 1) + 2 * tx, ix = (ox + 1 - kx) / 2;
        if ((ox + 1 - kx) < 0 || ix >= W) continue;
        const f32x4 v = PwcElem<TI>::ld4(in + ((size_t)(n * H + iy) * W + ix) * in_cs + in_co);
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(w + ((size_t)(ky * 4 + kx) * 2 + 0) * 4);
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(w + ((size_t)(ky * 4 + kx) * 2 + 1) * 4);
        a0 += v.x * k0.x + v.y * k0.y + v.z * k0.z + v.w * k0.w;
        a1 += v.x * k1.x + v.y * k1.y + v.z * k1.z + v.w * k1.w;
      }
    }
    TO* o = out + i * out_cs + out_co;
    if (pad4) PwcElem<TO>::st4(o, f32x4{a0 + bias[0], a1 + bias[1], 0.f, 0.f});
    else { o[0] = (TO)(a0 + bias[0]); o[1] = (TO)(a1 + bias[1]); }
  }
}

// The transpose conv of a WIDE tensor (up_feat: 565 -> 2 channels) in two steps, fp16 engine: the 16 taps x 2 outputs are 32 output
// channels of a 1x1 convolution, P[pixel][tap * 2 + o] = sum_c in[pixel][c] * kernel[tap][o][c] -- run as a 3x3 convolution whose
// only non-zero tap is the centre on the LDS-DMA kernel (conv3x3_dma.h), so the matrix pipe shares every weight fragment among 32
// pixels -- and this kernel gathers the four taps that reach an output pixel: out[2i + k - 1] += P[i][k].  (Eight lanes per output
// pixel re-read the 78 KB of weights from L2 for every pixel: 4.1 ms on the 608-channel level-2 buffer of a 5-frame stack.)
template <typename TE, bool PLANAR = false>      // PLANAR: P [tap][N, H, W][2] (the fp32 pointwise kernel) instead of [N, H, W][32]
__global__ void pwc_deconv_combine_kernel(const TE* __restrict__ P, const float* __restrict__ bias, TE* __restrict__ out, int out_cs, int out_co,
                                          int N, int H, int W, int pad4) {
  const int OH = 2 * H, OW = 2 * W;
  const size_t total = (size_t)N * OH * OW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), n = (int)(i / ((size_t)OW * OH));
    float a0 = bias[0], a1 = bias[1];
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) {
      const int ky = ((oy + 1) & 1) + 2 * ty, iy = (oy + 1 - ky) / 2;
      if ((oy + 1 - ky) < 0 || iy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int kx = ((ox + 1) & 1) + 2 * tx, ix = (ox + 1 - kx) / 2;
        if ((ox + 1 - kx) < 0 || ix >= W) continue;
        const TE* q = PLANAR ? P + ((size_t)(ky * 4 + kx) * N * H * W + (size_t)(n * H + iy) * W + ix) * 2
                             : P + ((size_t)(n * H + iy) * W + ix) * 32 + (ky * 4 + kx) * 2;
        a0 += (float)q[0]; a1 += (float)q[1];
      }
    }
    TE* o = out + i * out_cs + out_co;
    if (pad4) PwcElem<TE>::st4(o, f32x4{a0, a1, 0.f, 0.f});
    else { o[0] = (TE)a0; o[1] = (TE)a1; }
  }
}

// ---- fp32 engine: the layers with TWO output channels (the flow heads predict_flow/flow{l}, the context network's dc_conv{l}7, the
// up_feat deconvolutions) as a pointwise map + a gather ----
// out[p][o] = sum_tap sum_c x[p + tap][c] w[tap][c][o] = sum_tap T[p + tap][tap][o] with T[q][tap][o] = sum_c x[q][c] w[tap][c][o]:
// T is a 1x1 convolution to 9 x 2 (3x3 conv) or 16 x 2 (4x4 stride-2 deconvolution) channels -- it reads every input pixel exactly ONCE,
// in whole 128-byte pieces, no halo -- and the taps are gathered from the 80- / 128-byte records of T by a second, tiny kernel.  These
// layers read 450 - 670 channels per pixel to produce 2: they are bound by that read (9.6 GB for the level-2 flow head of a 5-frame
// stack), which the 16-row matrix kernel (14 of 16 MFMA rows padding: compute-bound on padded work, 5.2 ms), the eight-lanes-per-
// output-pixel deconvolution (4.3 ms for 2.4 GB) and the 64-wide generic kernel (dc_conv7: 2.3 ms for 0.5 GB) all missed by 2.5 - 8 x.
// Arithmetic: T^T [32 x pixels] = W^T [32 x Cin] * X^T [Cin x pixels] on v_mfma_f32_32x32x2_f32 -- the weights are the ROW operand (one
// register per K step, fetched once per 32-channel chunk and wave with four coalesced 16-byte loads of a host-packed array), the pixels
// the columns: lane (n, half) reads the 16-byte pieces 2 s + half of its own pixel's 144-byte LDS record (conflict-free) and holds 16
// sums of that pixel, four consecutive output channels per register quad -> 16-byte stores.  (A first version on the vector ALU with
// the weights in scalar registers, as head_conv.h has them, ran 25 k cycles per chunk: 46 - 74 KB of weights do not fit the 16 KB
// scalar cache, every s_load went to L2 with a handful in flight.)
struct PointwiseArgs {
  const float* in; int in_cs, in_co, Cin;    // channels [in_co, in_co + Cin) of an [npix, in_cs] buffer; Cin % 32 == 0, in_co % 4 == 0, in_cs % 4 == 0
  const float* w;                            // pack_pointwise(): [Cin / 32][4][64 lanes][4]
  float* out;                                // [NJ / 2 taps][npix][2]
  size_t npix;
};
constexpr int PW_PX = 256, PW_CH = 32, PW_REC = PW_CH * 4 + 16;

// w[c][j], j < nj <= 32 -> the kernel's row-operand order: K step (chunk, s, e) pairs channel chunk * 32 + 8 s + 4 half + e with lane (j, half)
inline void pack_pointwise(const float* w, int cin, int nj, std::vector<float>& out) {
  out.assign((size_t)cin * 32, 0.f);
  for (int c = 0; c < cin; ++c)
    for (int j = 0; j < nj; ++j) {
      const int chunk = c / 32, cc = c % 32, s_ = cc / 8, half = (cc % 8) / 4, e = cc % 4;
      out[(((size_t)chunk * 4 + s_) * 64 + half * 32 + j) * 4 + e] = w[(size_t)c * nj + j];
    }
}

template <int NJ>      // stored channels per pixel: 20 (3x3: 9 taps x 2, 2 padding) or 32 (4x4 deconvolution: 16 taps x 2)
__global__ __launch_bounds__(256) void pwc_pointwise_f32_kernel(const PointwiseArgs p) {
  __shared__ __attribute__((aligned(16))) char hs[PW_PX * PW_REC];
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, half = lane >> 5;
  const size_t p0 = (size_t)blockIdx.x * PW_PX;
  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[g][k] = 0.f;
  // loader: unit u = tid + 256 * i -> pixel u / 8 of the tile, 16-byte slot u % 8: eight consecutive lanes fetch one 128-byte piece
  constexpr int NU = PW_PX * (PW_CH / 4) / 256;
  const int slot = tid & 7;
  const float* src[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const size_t px = p0 + (tid >> 3) + 32 * i;
    src[i] = px < p.npix ? p.in + px * (size_t)p.in_cs + p.in_co + 4 * slot : nullptr;
  }
  f32x4 r[NU], wr[4];
  auto load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (src[i]) r[i] = *reinterpret_cast<const f32x4*>(src[i] + c0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) wr[s_] = *reinterpret_cast<const f32x4*>(p.w + (((size_t)(c0 / PW_CH) * 4 + s_) * 64 + lane) * 4);
  };
  load(0);
  for (int c0 = 0; c0 < p.Cin; c0 += PW_CH) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NU; ++i) *reinterpret_cast<f32x4*>(hs + ((tid >> 3) + 32 * i) * PW_REC + slot * 16) = r[i];
    f32x4 wc[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) wc[s_] = wr[s_];
    __syncthreads();
    if (c0 + PW_CH < p.Cin) load(c0 + PW_CH);               // the next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const char* rec = hs + (wave * 64 + g * 32 + n) * PW_REC + half * 16;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(rec + s_ * 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[s_][e], x[e], acc[g], 0, 0, 0);
      }
    }
  }
  // D[i][j]: column j = lane % 32 (the pixel), rows i = 8 (k / 4) + 4 half + k % 4 in register k.  Stored PLANAR -- out[i / 2][pixel][i % 2],
  // one plane of output pairs per tap -- so that 32 lanes write 256 contiguous bytes and the gather kernels read whole lines
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const size_t px = p0 + wave * 64 + g * 32 + n;
    if (px >= p.npix) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int tap = (8 * q + 4 * half) / 2 + t2;
        if (2 * tap < NJ) *reinterpret_cast<f2*>(p.out + ((size_t)tap * p.npix + px) * 2) = f2{acc[g][4 * q + 2 * t2], acc[g][4 * q + 2 * t2 + 1]};
      }
  }
}

// the nine taps of a 3x3 'same' convolution to two channels, gathered from the planes T [tap = ky * 3 + kx][N, H, W][2] (a tenth plane is
// padding), + bias (+ add): tf.layers.conv2d without activation (model_pwcnet.py:1447, :1519-1521)
__global__ void pwc_conv3_combine_kernel(const float* __restrict__ T, const float* __restrict__ bias, const float* __restrict__ add, int add_cs,
                                         int add_co, float* __restrict__ out, int out_cs, int out_co, int N, int H, int W) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const size_t total = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const f2 q = *reinterpret_cast<const f2*>(T + ((size_t)(ky * 3 + kx) * total + i + (size_t)((ky - 1) * W + (kx - 1))) * 2);
        a0 += q.x; a1 += q.y;
      }
    }
    a0 += bias[0]; a1 += bias[1];
    if (add) { a0 += add[i * add_cs + add_co]; a1 += add[i * add_cs + add_co + 1]; }
    out[i * out_cs + out_co] = a0; out[i * out_cs + out_co + 1] = a1;
  }
}

// (The padding channels of a decoder buffer -- 7 / 15 behind the cost volume, 2 behind up_flow and up_feat -- are written as zeros by
// the kernels that fill the group in front of them (r05; a launch of their own before that: 0.5 ms per stack, the level-2 ones 0.1-0.2 ms
// each for 4-14 bytes per 1152-byte pixel).  Everything else is written before it is read: no memset of the buffers.)

// conv1a of the feature pyramid (model_pwcnet.py:1092: 3 -> 16 channels, stride 2, 'same' = pad (0, 1) on the even sizes the
// network runs on, leaky relu): [N, H, W, 4] -> [N, H/2, W/2, 16] on v_mfma_f32_16x16x4_f32 -- rows = the 16 output channels, columns =
// 16 output pixels, K = the 4 channels of ONE input pixel, so a tap is one MFMA whose row operand (w[tap][k][o] at lane o + 16 k) is
// loaded once per wave and whose column operand is channel lane / 16 of input pixel (2 ox + kx, 2 oy + ky): nine 4-byte loads and nine
// MFMAs per 16 output pixels, the 16 sums of a pixel in four lanes' register quads -> 16-byte stores.  Weights [9][4][16] + bias [16]
// by value.  (r03's vector-ALU version took 1.0 ms per 5-frame stack: the compiler hoisted the 592 scalar weight loads out of the
// pixel loop and spilled them to vector lanes -- 3057 v_readlane per pixel for 150 packed FMAs; with the weights in LDS 0.82 ms, bound
// by 108 uniform 16-byte LDS reads per pixel.)
struct Conv1aWeights { float w[9 * 64]; float bias[16]; };      // by value: kernel-argument memory
template <typename TE>
__global__ __launch_bounds__(256) void pwc_conv1a_kernel(const TE* __restrict__ in, const Conv1aWeights cw, TE* __restrict__ out, int N, int H,
                                                         int W, float slope) {
  const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
  float wa[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wa[t] = cw.w[t * 64 + kq * 16 + col];          // row operand: lane (o = col, k = kq)
  f32x4 b4;
#pragma unroll
  for (int e = 0; e < 4; ++e) b4[e] = cw.bias[4 * kq + e];                   // result register e of lane (pixel col, quad kq) = output 4 kq + e
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * OH * OW, groups = (total + 15) / 16;
  const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t g = wave0; g < groups; g += nwaves) {
    const size_t i = g * 16 + col;
    const bool live = i < total;
    const size_t ii = live ? i : total - 1;
    const int ox = (int)(ii % OW), oy = (int)((ii / OW) % OH), n = (int)(ii / ((size_t)OW * OH));
    float x[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy + ky, ix = 2 * ox + kx;
        x[ky * 3 + kx] = (iy < H && ix < W) ? (float)in[((size_t)(n * H + iy) * W + ix) * 4 + kq] : 0.f;
      }
    f32x4 acc = b4;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t], x[t], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] >= 0.f ? acc[e] : acc[e] * slope;
    if (live) PwcElem<TE>::st4(out + i * 16 + 4 * kq, acc);
  }
}

// The fp16 engine's conv1a (r05): K = 16 of v_mfma_f32_16x16x16_f16 is FOUR taps x the 4 channels of an input pixel, so lane (pixel
// col, quad kq) fetches the whole 8-byte pixel of tap 4 m + kq for MFMA m: three 8-byte loads and three MFMAs per 16 output pixels
// instead of nine 2-byte loads and nine fp32 MFMAs (497 us per 5-frame stack at 1.35 TB/s).  Weights rounded to fp16 like every other
// layer of this engine; bias, accumulation and the leaky relu in fp32.
__global__ __launch_bounds__(256) void pwc_conv1a_f16_kernel(const _Float16* __restrict__ in, const Conv1aWeights cw, _Float16* __restrict__ out,
                                                             int N, int H, int W, float slope) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
  h4 wa[3];
  int dy[3], dx[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int t = 4 * m + kq;                                                  // this lane's tap of MFMA m (taps 9 .. 11 do not exist)
#pragma unroll
    for (int c = 0; c < 4; ++c) wa[m][c] = t < 9 ? (_Float16)cw.w[t * 64 + c * 16 + col] : (_Float16)0.f;
    dy[m] = t < 9 ? t / 3 : 0; dx[m] = t < 9 ? t % 3 : 0;
  }
  f32x4 b4;
#pragma unroll
  for (int e = 0; e < 4; ++e) b4[e] = cw.bias[4 * kq + e];                     // result register e of lane (pixel col, quad kq) = output 4 kq + e
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * OH * OW, groups = (total + 15) / 16;
  const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t g = wave0; g < groups; g += nwaves) {
    const size_t i = g * 16 + col;
    const bool live = i < total;
    const size_t ii = live ? i : total - 1;
    const int ox = (int)(ii % OW), oy = (int)((ii / OW) % OH), n = (int)(ii / ((size_t)OW * OH));
    h4 x[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int iy = 2 * oy + dy[m], ix = 2 * ox + dx[m];
      x[m] = h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      if (iy < H && ix < W && 4 * m + kq < 9) x[m] = *reinterpret_cast<const h4*>(in + ((size_t)(n * H + iy) * W + ix) * 4);
    }
    f32x4 acc = b4;
#pragma unroll
    for (int m = 0; m < 3; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wa[m], x[m], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] >= 0.f ? acc[e] : acc[e] * slope;
    if (live) PwcElem<_Float16>::st4(out + i * 16 + 4 * kq, acc);
  }
}

// cost volume: out[px][(dy+4)*9 + (dx+4)] = leaky_relu(mean_c c1[px][c] * w2[px + (dy, dx)][c], 0.1), zero outside.
// One workgroup per 8x32 pixel tile, one thread per pixel with all 81 sums in registers; the (8+8) x (32+8) halo of w2
// goes through LDS 16 channels at a time (80-byte records: a ds_read_b128 phase of 16 neighbouring pixels hits every
// bank once), so w2 is read from HBM/L2 once per tile instead of 81 times per pixel through L1.
constexpr int CV_CH = 16;
constexpr int CV_REC = CV_CH * 4 + 16;
constexpr int CV_HW = TILE_W + 8, CV_HH = TILE_H + 8;
constexpr size_t costvol_lds_bytes() { return (size_t)CV_HH * CV_HW * CV_REC; }
// fp16 engine (r05): the halo chunk stays fp16 in LDS -- 32 bytes of a 48-byte record (12 banks apart: a ds_read_b128 phase of 16
// neighbouring pixels still hits every bank once) -- and a displacement's 16 products are eight v_dot2_f32_f16 (two fp16 products + an
// fp32 accumulator each).  The fp32 form of this kernel is bound by its LDS reads (324 ds_read_b128 per pixel and chunk: twice the
// cycles of its 1296 FMAs); halving both took level 2 of a 5-frame stack from 944 us to the figure in DESIGN 3.4.
constexpr int CV_REC16 = CV_CH * 2 + 16;
typedef _Float16 cv_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 cv_h4 __attribute__((ext_vector_type(4)));

// Batched form: image n of c1 is c1 + (size_t)c1_img[n] * H * W * c1_cs (the features of the item's first frame, or its slot of the
// decoder buffer), likewise w2.
template <typename TE>
__global__ __launch_bounds__(256) void pwc_costvol_kernel(const TE* __restrict__ c1, int c1_cs, int c1_co, const PwcItems c1_img,
                                                          const TE* __restrict__ w2, const PwcItems w2_img, int C,
                                                          TE* __restrict__ out, int out_cs, int out_co, int N, int H, int W, int zero_pad) {
  extern __shared__ __attribute__((aligned(16))) char cv_smem[];
  const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  int t = blockIdx.x;
  const int tx_ = t % tiles_x; t /= tiles_x;
  const int ty_ = t % tiles_y;
  const int n = t / tiles_y;
  const int x0 = tx_ * TILE_W, y0 = ty_ * TILE_H;
  const int x = x0 + tx, y = y0 + ty;
  const bool inside = x < W && y < H;
  const size_t pix_in = (size_t)min(y, H - 1) * W + min(x, W - 1);
  const size_t pix = (size_t)n * H * W + pix_in;
  const TE* c1n = c1 + (size_t)c1_img.a[n] * H * W * c1_cs + c1_co;
  const TE* w2n = w2 + (size_t)w2_img.a[n] * H * W * C;
  float s[81];
#pragma unroll
  for (int k = 0; k < 81; ++k) s[k] = 0.f;
  if constexpr (std::is_same<TE, _Float16>::value) {
    for (int c0 = 0; c0 < C; c0 += CV_CH) {
      const int nq = min(4, (C - c0) >> 2);               // 4-channel quarters of this chunk that exist (C % 4 == 0)
      for (int i = tid; i < CV_HH * CV_HW * 4; i += 256) {
        const int hp = i >> 2, q = i & 3;
        const int hy = hp / CV_HW, hx = hp - hy * CV_HW;
        const int gy = y0 - 4 + hy, gx = x0 - 4 + hx;
        uint2 v = make_uint2(0u, 0u);
        if (q < nq && gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = *reinterpret_cast<const uint2*>(w2n + ((size_t)gy * W + gx) * C + c0 + 4 * q);
        *reinterpret_cast<uint2*>(cv_smem + hp * CV_REC16 + q * 8) = v;
      }
      cv_h2 a[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cv_h4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (q < nq) v = *reinterpret_cast<const cv_h4*>(c1n + pix_in * c1_cs + c0 + 4 * q);
        a[2 * q] = cv_h2{v.x, v.y}; a[2 * q + 1] = cv_h2{v.z, v.w};
      }
      __syncthreads();
#pragma unroll
      for (int dy = 0; dy < 9; ++dy)
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
          const char* r = cv_smem + ((ty + dy) * CV_HW + tx + dx) * CV_REC16;
          float acc = s[dy * 9 + dx];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint4 u = *reinterpret_cast<const uint4*>(r + q * 16);
            acc = __builtin_amdgcn_fdot2(a[4 * q], __builtin_bit_cast(cv_h2, u.x), acc, false);
            acc = __builtin_amdgcn_fdot2(a[4 * q + 1], __builtin_bit_cast(cv_h2, u.y), acc, false);
            acc = __builtin_amdgcn_fdot2(a[4 * q + 2], __builtin_bit_cast(cv_h2, u.z), acc, false);
            acc = __builtin_amdgcn_fdot2(a[4 * q + 3], __builtin_bit_cast(cv_h2, u.w), acc, false);
          }
          s[dy * 9 + dx] = acc;
        }
      __syncthreads();
    }
  } else
  for (int c0 = 0; c0 < C; c0 += CV_CH) {
    const int nq = min(4, (C - c0) >> 2);                 // 16-byte quarters of this chunk that exist (C % 4 == 0)
    for (int i = tid; i < CV_HH * CV_HW * 4; i += 256) {  // halo chunk of w2 -> LDS, zeros outside the image
      const int hp = i >> 2, q = i & 3;
      const int hy = hp / CV_HW, hx = hp - hy * CV_HW;
      const int gy = y0 - 4 + hy, gx = x0 - 4 + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < nq && gy >= 0 && gy < H && gx >= 0 && gx < W)
        v = PwcElem<TE>::ld4(w2n + ((size_t)gy * W + gx) * C + c0 + 4 * q);
      *reinterpret_cast<f32x4*>(cv_smem + hp * CV_REC + q * 16) = v;
    }
    f32x4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < nq) a[q] = PwcElem<TE>::ld4(c1n + pix_in * c1_cs + c0 + 4 * q);
    }
    __syncthreads();
#pragma unroll
    for (int dy = 0; dy < 9; ++dy)
#pragma unroll
      for (int dx = 0; dx < 9; ++dx) {
        const char* r = cv_smem + ((ty + dy) * CV_HW + tx + dx) * CV_REC;
        float acc = s[dy * 9 + dx];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(r + q * 16);
          acc += a[q].x * b.x + a[q].y * b.y + a[q].z * b.z + a[q].w * b.w;
        }
        s[dy * 9 + dx] = acc;
      }
    __syncthreads();
  }
  if (!inside) return;
  const float inv = 1.f / (float)C;
  TE* o = out + pix * out_cs + out_co;
#pragma unroll
  for (int k = 0; k < 80; k += 4) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float m = s[k + e] * inv; v[e] = m >= 0.f ? m : 0.1f * m; }
    PwcElem<TE>::st4(o + k, v);
  }
  const float m80 = s[80] * inv, v80 = m80 >= 0.f ? m80 : 0.1f * m80;
  if (zero_pad == 0) { o[80] = (TE)v80; return; }
  // zero_pad (a multiple of 4, the decoder's 7 or 15): the padding channels behind the 81 displacements read as zeros in the next
  // convolution; written here they are part of the pixel's one contiguous run (as a launch of their own they were 184 us at level 2)
  PwcElem<TE>::st4(o + 80, f32x4{v80, 0.f, 0.f, 0.f});
  for (int k = 84; k < 81 + zero_pad; k += 4) PwcElem<TE>::st4(o + k, f32x4{0.f, 0.f, 0.f, 0.f});
}

// dense_image_warp: out[px][c] = bilinear(img, x + scale*u, y + scale*v); floor index clamped to [0, size-2], weight
// clamped to [0, 1] (tf.contrib.image); evaluation order top = ax*(tr - tl) + tl, ... as tf.contrib's.
// Batched form: image n samples img + (size_t)img_idx.a[n] * H * W * C.
template <typename TE>
__global__ void pwc_warp_kernel(const TE* __restrict__ img, const PwcItems img_idx, int C, const TE* __restrict__ flow, int f_cs, int f_co,
                                float scale, TE* __restrict__ out, int N, int H, int W) {
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const size_t pix = i / C4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
    const float qx = (float)x + (float)flow[pix * f_cs + f_co] * scale, qy = (float)y + (float)flow[pix * f_cs + f_co + 1] * scale;
    const float fx = fminf(fmaxf(floorf(qx), 0.f), (float)(W - 2)), fy = fminf(fmaxf(floorf(qy), 0.f), (float)(H - 2));
    const float ax = fminf(fmaxf(qx - fx, 0.f), 1.f), ay = fminf(fmaxf(qy - fy, 0.f), 1.f);
    const int x0 = (int)fx, y0 = (int)fy;
    const TE* b = img + (size_t)img_idx.a[n] * H * W * C + 4 * c;
    const f32x4 tl = PwcElem<TE>::ld4(b + ((size_t)y0 * W + x0) * C), tr = PwcElem<TE>::ld4(b + ((size_t)y0 * W + x0 + 1) * C);
    const f32x4 bl = PwcElem<TE>::ld4(b + ((size_t)(y0 + 1) * W + x0) * C), br = PwcElem<TE>::ld4(b + ((size_t)(y0 + 1) * W + x0 + 1) * C);
    const f32x4 top = ax * (tr - tl) + tl, bot = ax * (br - bl) + bl;
    PwcElem<TE>::st4(out + 4 * i, ay * (bot - top) + top);
  }
}

// copy a channel range (feature level c1 into the decoder's concatenated buffer)
// (batched: item n of dst takes image src_idx.a[n] of src; npix = pixels per image)
template <typename TE>
__global__ void pwc_copy_channels_kernel(const TE* __restrict__ src, const PwcItems src_idx, int C, TE* __restrict__ dst, int d_cs, int d_co,
                                         size_t npix, int N) {
  const int C4 = C / 4;
  const size_t per = npix * C4, total = per * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per);
    const size_t r = i - (size_t)n * per;
    const size_t pix = r / C4;
    const int c = (int)(r % C4);
    PwcElem<TE>::st4(dst + ((size_t)n * npix + pix) * d_cs + d_co + 4 * c, PwcElem<TE>::ld4(src + ((size_t)src_idx.a[n] * npix + pix) * C + 4 * c));
  }
}

// Pre-processing of one frame: YUV uint8 [h, w, 3] -> network input [PH, PW, 4] (RGB / 255, channel 3 = 0, zero padding
// below / right of the 2h x 2w image).  Double maths up to the uint8 truncation, as the reference's numpy / skimage do.
template <typename TE>
__global__ void pwc_prep_kernel(const uint8_t* __restrict__ yuv, int h, int w, TE* __restrict__ out, int PH, int PW,
                                const ColorConsts cc) {
#pragma clang fp contract(off)
  const size_t total = (size_t)PH * PW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % PW), Y = (int)(i / PW);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (Y < 2 * h && X < 2 * w) {
      // skimage resize x2, order 1, half-pixel centres, mode 'reflect' (= ndimage 'mirror'): even outputs take
      // 0.25*in[i-1] + 0.75*in[i], odd ones 0.75*in[i] + 0.25*in[i+1]; in[-1] = in[1], in[n] = in[n-2]; rows first.
      const int yi = Y >> 1, xi = X >> 1;
      int ya = (Y & 1) ? yi : yi - 1, yb = (Y & 1) ? yi + 1 : yi;      // lower / upper source rows
      int xa = (X & 1) ? xi : xi - 1, xb = (X & 1) ? xi + 1 : xi;
      const double wya = (Y & 1) ? 0.75 : 0.25, wxa = (X & 1) ? 0.75 : 0.25;
      if (ya < 0) ya = 1; if (yb >= h) yb = h - 2;
      if (xa < 0) xa = 1; if (xb >= w) xb = w - 2;
      double p[2][2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint8_t* s = yuv + ((size_t)(r ? yb : ya) * w + (c ? xb : xa)) * 3;
          const float f[3] = {(float)s[0], (float)s[1], (float)s[2]};
          yuv2rgb_d(cc, f, p[r][c]);
        }
      float rgb[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // rows (axis 0) first, then columns, as the restated skimage passes do
        const double ca = wya * p[0][0][k] + (1.0 - wya) * p[1][0][k];
        const double cb = wya * p[0][1][k] + (1.0 - wya) * p[1][1][k];
        const double v = wxa * ca + (1.0 - wxa) * cb;
        rgb[k] = (float)(uint8_t)v / 255.f;                          // np.array(.., dtype=uint8) truncation; adapt_x / 255
      }
      o.x = rgb[0]; o.y = rgb[1]; o.z = rgb[2];
    }
    PwcElem<TE>::st4(out + 4 * i, o);        // (k / 255, k = 0 .. 255: fp16 holds it to 2^-12 relative)
  }
}

// Post-processing: flow2 [FH, FW] (level-2 flow, channels at f_co of a buffer with pixel stride f_cs) -> out [h, w, 2]:
// flow_pred = legacy bilinear x4 * 4 (model_pwcnet.py:1587-1590), cropped to 2h x 2w, skimage anti-aliased resize to
// h x w (Gaussian sigma 0.5 radius 2 'mirror' on both axes, then the mean of samples 2i, 2i+1), / 2 (script :139).
__global__ void pwc_flow_out_kernel(const float* __restrict__ f2, int f_cs, int f_co, int FH, int FW, float* __restrict__ out,
                                    int h, int w) {
  // effective 6-tap kernel of "Gaussian then average of two neighbours": e[t] = (g[t] + g[t-1]) / 2, t = -2..3
  const double gk[5] = {3.3546262790251185e-04, 1.3533528323661270e-01, 1.0, 1.3533528323661270e-01, 3.3546262790251185e-04};
  const double gs = gk[0] + gk[1] + gk[2] + gk[3] + gk[4];
  float e[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) e[t] = (float)(0.5 * ((t < 5 ? gk[t] : 0.0) + (t > 0 ? gk[t - 1] : 0.0)) / gs);
  const int H2 = 2 * h, W2 = 2 * w;
  const size_t total = (size_t)h * w;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)(i / w);
    // Both filters are separable and the six x2-frame rows 2y-2 .. 2y+3 (mirrored) fall into at most three level-2 rows (+ one
    // for the bilinear partner): filter those four rows horizontally once (96 loads instead of 288), then combine vertically.
    int Yt[6], rmin = 1 << 30;
#pragma unroll
    for (int ty = 0; ty < 6; ++ty) {
      int Y = 2 * y + ty - 2;
      Y = Y < 0 ? -Y : (Y >= H2 ? 2 * H2 - 2 - Y : Y);
      Yt[ty] = Y;
      rmin = min(rmin, Y >> 2);
    }
    int x0t[6], x1t[6];
    float fxt[6];
#pragma unroll
    for (int tx = 0; tx < 6; ++tx) {
      int X = 2 * x + tx - 2;
      X = X < 0 ? -X : (X >= W2 ? 2 * W2 - 2 - X : X);
      x0t[tx] = X >> 2; x1t[tx] = min(x0t[tx] + 1, FW - 1);
      fxt[tx] = (float)(X & 3) * 0.25f;
    }
    float hr[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* row = f2 + (size_t)min(rmin + r, FH - 1) * FW * f_cs + f_co;
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int tx = 0; tx < 6; ++tx) {
        const float* l = row + (size_t)x0t[tx] * f_cs; const float* rr = row + (size_t)x1t[tx] * f_cs;
        r0 += e[tx] * (l[0] + (rr[0] - l[0]) * fxt[tx]);
        r1 += e[tx] * (l[1] + (rr[1] - l[1]) * fxt[tx]);
      }
      hr[r][0] = r0; hr[r][1] = r1;
    }
    auto pick = [&](int idx, int k) { return idx == 0 ? hr[0][k] : (idx == 1 ? hr[1][k] : (idx == 2 ? hr[2][k] : hr[3][k])); };
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ty = 0; ty < 6; ++ty) {
      const int y0 = Yt[ty] >> 2, y1 = min(y0 + 1, FH - 1);
      const float fy = (float)(Yt[ty] & 3) * 0.25f;
      const float t0 = pick(y0 - rmin, 0), b0 = pick(y1 - rmin, 0), t1 = pick(y0 - rmin, 1), b1 = pick(y1 - rmin, 1);
      a0 += e[ty] * (t0 + (b0 - t0) * fy);
      a1 += e[ty] * (t1 + (b1 - t1) * fy);
    }
    out[i * 2] = a0 * 2.f;                 // x 4 (model_pwcnet.py:1590) / 2 (script :139)
    out[i * 2 + 1] = a1 * 2.f;
  }
}

// flow buffer (pixel stride 4) -> dense [px, 2]
__global__ void stitch_free_copy2_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    dst[2 * i] = src[4 * i];
    dst[2 * i + 1] = src[4 * i + 1];
  }
}

// flow_pred = tf.image.resize_bilinear(flow2, x4) * 4 (model_pwcnet.py:1587-1590), TF-1.13 legacy kernel
__global__ void pwc_upsample4_kernel(const float* __restrict__ f2, int f_cs, int f_co, int FH, int FW, float* __restrict__ out) {
  const int H = 4 * FH, W = 4 * FW;
  const size_t total = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int X = (int)(i % W), Y = (int)(i / W);
    const float sy = (float)Y * 0.25f, sx = (float)X * 0.25f;
    const int y0 = (int)sy, y1 = min(y0 + 1, FH - 1), x0 = (int)sx, x1 = min(x0 + 1, FW - 1);
    const float fy = sy - (float)y0, fx = sx - (float)x0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float tl = f2[((size_t)y0 * FW + x0) * f_cs + f_co + k], tr = f2[((size_t)y0 * FW + x1) * f_cs + f_co + k];
      const float bl = f2[((size_t)y1 * FW + x0) * f_cs + f_co + k], br = f2[((size_t)y1 * FW + x1) * f_cs + f_co + k];
      const float top = tl + (tr - tl) * fx, bot = bl + (br - bl) * fx;
      out[i * 2 + k] = (top + (bot - top) * fy) * 4.f;
    }
  }
}

}  // namespace fisr
