// 3x3 SAME convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces every `tf.nn.conv2d(x, w, [1,1,1,1], 'SAME') + b` of the reference
// (ops.py:7-11) together with the element-wise work around it:
//   relu on the input        (ops.py:41-42  Conv2d(relu(x)))
//   relu on the output       (ops.py:52,62,70,75)
//   residual add             (ops.py:43  n = x + n)
//   channel concat of 2 srcs (ops.py:71  tf.concat([n, skip], 3))
//   depth_to_space(.,2)      (FISRnet.py:99, NHWC "DCR" order) folded into the store
//   channel scatter into the 9-ch prediction (FISRnet.py:107-108 split/concat)
//
// GEMM view: M = output pixels, N = output channels, K = 9 taps x Cin.
// Workgroup = 256 threads = 4 wave64; output tile = 8 rows x 32 cols x BN channels
// (BN = 32*NT).  Each wave owns 2 image rows (2 M-subtiles of 32 pixels) x NT
// N-subtiles of 32 channels = 2*NT accumulators of the 32x32 MFMA (16 VGPR each).
// The K loop walks the input channels in 64-byte chunks (16 fp32 / 32 fp16 channels):
// the (8+2)x(32+2) halo tile of the chunk and the 9 x BN x chunk weights are staged in
// LDS once, then all 9 taps read shifted windows of the same halo tile (9x LDS reuse of
// every input byte).  One ds_read_b128 per lane delivers a 16-byte k-slice that feeds
// 4 v_mfma_f32_32x32x2_f32 (fp32) or 1 v_mfma_f32_32x32x16_f16 (fp16).
//
// LDS records are 64 data bytes + 16 pad = 80 B (5 x 16-B slots, odd) so that the 16
// lanes of every ds_read_b128 service group (which always cover all residues mod 16 of
// the pixel / channel index) land on 16 distinct 16-B slots: conflict-free
// (MI355X_MICROARCH.md, LDS table).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fisr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int TILE_H = 8;
constexpr int TILE_W = 32;
constexpr int HALO_W = TILE_W + 2;
constexpr int HALO_PIX = (TILE_H + 2) * HALO_W;  // 340
constexpr int CHUNK_BYTES = 64;                  // channel bytes staged per K chunk
constexpr int REC_BYTES = CHUNK_BYTES + 16;      // LDS record stride (odd number of 16-B slots)
constexpr int CONV_THREADS = 256;

template <typename T> struct Prec;

template <> struct Prec<float> {
  static constexpr int CC = CHUNK_BYTES / 4;  // 16 channels per chunk
  typedef f32x4 Frag;
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  static __device__ __forceinline__ uint4 relu16(uint4 v) {
    f32x4 f = __builtin_bit_cast(f32x4, v);
    f.x = fmaxf(f.x, 0.f); f.y = fmaxf(f.y, 0.f); f.z = fmaxf(f.z, 0.f); f.w = fmaxf(f.w, 0.f);
    return __builtin_bit_cast(uint4, f);
  }
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};

template <> struct Prec<_Float16> {
  static constexpr int CC = CHUNK_BYTES / 2;  // 32 channels per chunk
  typedef f16x8 Frag;
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  static __device__ __forceinline__ uint4 relu16(uint4 v) {
    f16x8 f = __builtin_bit_cast(f16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f[i] > (_Float16)0 ? f[i] : (_Float16)0;
    return __builtin_bit_cast(uint4, f);
  }
  static __device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
  static __device__ __forceinline__ _Float16 from_f32(float v) { return (_Float16)v; }
};

struct ConvArgs {
  const void* in0;   // [N,H,W,C0]
  const void* in1;   // [N,H,W,C1] second concat source (nullable)
  const void* wpk;   // packed weights [Cin/CC][9][CoutPad][CC]
  const float* bias; // [CoutPad]
  const void* res;   // residual [N,H,W,Cout] (nullable; may alias out)
  void* out;
  int C0, C1;        // multiples of CC
  int N, H, W;
  int Cout, CoutPad;
  int relu_in, relu_out, d2s;
  int d2s_shift;     // log2(Cout/4) when d2s (Cout/4 must be a power of two)
  // channel scatter of the store: oc = n + coff + (n >= split ? gap : 0), row stride cstride
  int out_cstride, out_coff, out_split, out_gap;
};

template <typename T, int NT> constexpr size_t conv_lds_bytes() {
  return (size_t)HALO_PIX * REC_BYTES + (size_t)9 * 32 * NT * REC_BYTES;
}

template <typename T, int NT, bool OUT_F32>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv3x3_mfma_kernel(const ConvArgs p) {
  typedef Prec<T> P;
  typedef typename P::Frag Frag;
  constexpr int CC = P::CC;
  constexpr int BN = 32 * NT;
  constexpr int EPU = 16 / sizeof(T);  // elements per 16-byte unit

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_in = smem;
  char* s_w = smem + HALO_PIX * REC_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int kh = lane >> 5;

  const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
  const int tiles_y = (p.H + TILE_H - 1) / TILE_H;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int nb = t / tiles_y;
  const int x0 = tx * TILE_W, y0 = ty * TILE_H;
  const int n0 = blockIdx.y * BN;

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  // ---- loader geometry (identical for every K chunk) ----
  // A 16-byte unit u = tid + i*256 covers pixel (u>>2) of the halo tile, slot (u&3) of its
  // 64-byte chunk record; slot and the pixel's low part are per-thread constants.
  constexpr int NIN = (HALO_PIX * 4 + CONV_THREADS - 1) / CONV_THREADS;  // 6
  constexpr int NWT = (9 * BN * 4 + CONV_THREADS - 1) / CONV_THREADS;    // 9 (BN=64) / 5 (BN=32)
  const int slot = tid & 3;
  int in_pix[NIN];  // linear pixel index in the source image, -1 = zero padding / no unit
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int pix = (tid >> 2) + i * (CONV_THREADS / 4);
    const int py = pix / HALO_W, px = pix - py * HALO_W;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = pix < HALO_PIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    in_pix[i] = ok ? (nb * p.H + gy) * p.W + gx : -1;
  }
  uint4 rin[NIN], rwt[NWT];
  const char* a_base = s_in + ((wave * 2) * HALO_W + li) * REC_BYTES + kh * 16;
  const char* b_base = s_w + li * REC_BYTES + kh * 16;
  const int nchunks = (p.C0 + p.C1) / CC;

  // Software pipeline: iteration kc writes chunk kc (already in registers) to LDS, issues the
  // global loads of chunk kc+1 (they stay in flight during the MFMAs), then computes chunk kc.
  // Iteration -1 only issues the first loads.
  for (int kc = -1; kc < nchunks; ++kc) {
    if (kc >= 0) {
      __syncthreads();  // every wave is done reading the previous chunk from LDS
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        const int pix = (tid >> 2) + i * (CONV_THREADS / 4);
        if (i < NIN - 1 || pix < HALO_PIX)
          *reinterpret_cast<uint4*>(s_in + pix * REC_BYTES + slot * 16) = p.relu_in ? P::relu16(rin[i]) : rin[i];
      }
#pragma unroll
      for (int i = 0; i < NWT; ++i) {
        const int r = (tid >> 2) + i * (CONV_THREADS / 4);
        if (NWT * CONV_THREADS == 9 * BN * 4 || r < 9 * BN)
          *reinterpret_cast<uint4*>(s_w + r * REC_BYTES + slot * 16) = rwt[i];
      }
      __syncthreads();
    }
    if (kc + 1 < nchunks) {  // global -> registers for the next chunk
      const T* src;
      int csrc, coff;
      const int c0 = (kc + 1) * CC;
      if (c0 < p.C0) { src = (const T*)p.in0; csrc = p.C0; coff = c0; }
      else           { src = (const T*)p.in1; csrc = p.C1; coff = c0 - p.C0; }
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (in_pix[i] >= 0)
          v = *reinterpret_cast<const uint4*>(src + (size_t)in_pix[i] * csrc + coff + slot * EPU);
        rin[i] = v;
      }
      const T* wsrc = (const T*)p.wpk + (size_t)(kc + 1) * 9 * p.CoutPad * CC;
#pragma unroll
      for (int i = 0; i < NWT; ++i) {
        const int r = (tid >> 2) + i * (CONV_THREADS / 4);  // r = tap*BN + n
        const int tap = r / BN, n = r - tap * BN;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (NWT * CONV_THREADS == 9 * BN * 4 || r < 9 * BN)
          v = *reinterpret_cast<const uint4*>(wsrc + ((size_t)tap * p.CoutPad + n0 + n) * CC + slot * EPU);
        rwt[i] = v;
      }
    }
    if (kc < 0) continue;

    // ---- 9 taps x 2 k-groups of MFMA on the staged chunk ----
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        Frag a[2], b[NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
          a[m] = *reinterpret_cast<const Frag*>(a_base + ((m + dy) * HALO_W + dx) * REC_BYTES + kg * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          b[j] = *reinterpret_cast<const Frag*>(b_base + (tap * BN + j * 32) * REC_BYTES + kg * 32);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j) P::mma(acc[m][j], a[m], b[j]);
      }
    }
  }

  // ---- epilogue: bias, residual, relu, (pixel-shuffle / channel-scatter) store ----
  // C/D layout of the 32x32 MFMA: column (N) = lane & 31, row (M) = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Every store address is (per-(row,subtile) base) + (compile-time pixel offset) * step.
  const int xb = x0 + 4 * kh;        // first pixel column this lane holds
  const int xlim = p.W - xb;         // columns xb + q are valid for q < xlim
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 32 + li;
    if (n >= p.Cout) continue;
    const float bv = p.bias[n];
    int step;          // output elements between horizontally adjacent pixels
    int nmap, sub_y = 0, sub_x = 0;
    if (p.d2s) {
      const int sub = n >> p.d2s_shift;  // n / (Cout/4)
      nmap = n & ((1 << p.d2s_shift) - 1);
      sub_y = sub >> 1; sub_x = sub & 1;
      step = 2 << p.d2s_shift;
    } else {
      nmap = n + p.out_coff + (n >= p.out_split ? p.out_gap : 0);
      step = p.out_cstride;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y = y0 + wave * 2 + m;
      if (y >= p.H) continue;
      const size_t pix = (size_t)(nb * p.H + y) * p.W + xb;
      size_t obase;
      if (p.d2s) obase = (((size_t)(nb * 2 * p.H + 2 * y + sub_y)) * (2 * p.W) + 2 * xb + sub_x) * (size_t)(1 << p.d2s_shift) + nmap;
      else obase = pix * (size_t)step + nmap;
      const T* rp = p.res ? (const T*)p.res + pix * p.Cout + n : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        constexpr int dummy = 0; (void)dummy;
        const int q = (r & 3) + 8 * (r >> 2);
        if (q >= xlim) continue;
        float v = acc[m][j][r] + bv;
        if (rp) v += P::to_f32(rp[q * p.Cout]);
        if (p.relu_out) v = fmaxf(v, 0.f);
        if (OUT_F32) ((float*)p.out)[obase + (size_t)(q * step)] = v;
        else ((T*)p.out)[obase + (size_t)(q * step)] = P::from_f32(v);
      }
    }
  }
}

}  // namespace fisr
