// The 3- and 6-channel heads of the fp32 engine (FISRnet.py:100,105: Conv2d(relu(n2), [3,3,64,6]) and Conv2d(n3, [3,3,64,3]) on
// the pixel-shuffled 2H x 2W x 64 maps) on the VECTOR ALU.
//
// On the matrix pipe a head wastes most of every MFMA: 3 or 6 useful rows of the 16 the smallest fp32 MFMA has (the 16-row
// variant of conv3x3.h: 92 TF/s of padded work, 6 % of the fp32 step).  The work itself is small -- 9 x 64 x (3 | 6) FMAs
// per pixel -- and the packed fp32 FMA (v_pk_fma_f32: two lanes' worth per instruction) runs at the same 256 FLOP/clk/CU
// as the fp32 MFMA, so here every lane owns one output pixel and keeps its 3 | 6 sums in registers:
//     acc[o .. o+1] += x[tap][c] * w[tap][c][o .. o+1]            one v_pk_fma_f32 per channel and output pair,
// x from the LDS halo tile (a ds_read_b128 = 4 channels feeds 12 | 8 packed FMAs), the weights -- uniform over the
// workgroup -- straight from scalar registers (s_load), the op_sel bits broadcasting one x to both halves.
// Tile 8 x 32 pixels, 256 threads, 16 or 32 channels per K chunk (HeadCfg).
#pragma once
#include "conv3x3.h"

namespace fisr {

struct HeadArgs {
  const float* in;    // [N, H, W, Cin] fp32, Cin % 32 == 0
  const float* w;     // [9][Cin][2 * NPAIR]: outputs padded to the kernel's pairs (zeros)
  const float* bias;  // [8]
  float* out;         // channel n of pixel p at out[p * out_cstride + n + out_coff + (n >= out_split ? out_gap : 0)]
  int N, H, W, Cin, Cout, relu_in, relu_out;
  int out_cstride, out_coff, out_split, out_gap;
};

// K chunk: the kernel is bound by its read of the 64-channel 4K tensor (6.6 GB per launch at full size: 3.1 TB/s, its
// FMAs, LDS reads and scalar weight loads can all be ablated without changing the time), so the chunk size is what
// measured best per variant: 32 channels = one whole 128-byte line of every pixel record per pass (144-byte LDS records,
// 3 workgroups per CU) for the 3-channel head (2116 -> 1749 us), 16 channels (80-byte records, 5 workgroups per CU) for
// the 6-channel one, whose longer FMA phases need the extra workgroups (2161 vs 2473 us with 32).
// Both record sizes put the 16 lanes of a ds_read_b128 phase on 16 distinct bank groups.
#ifndef FISR_HEAD_TH
#define FISR_HEAD_TH 8            // tile height (A/B hook: 16 = 512 lanes, halo re-read 1.195 instead of 1.33, 16-channel chunks for both heads)
#endif
constexpr int HEAD_TH = FISR_HEAD_TH, HEAD_NTHR = 32 * HEAD_TH, HEAD_HALO_PIX = (HEAD_TH + 2) * HALO_W;
template <int NPAIR> struct HeadCfg {
  static constexpr int CH = (NPAIR == 2 && HEAD_TH == 8) ? 32 : 16;     // channels per K chunk
  static constexpr int REC = CH * 4 + 16;             // LDS bytes per pixel record
  static constexpr int SLOTS = CH / 4;                // 16-byte slots per record
};
constexpr int HEAD_CH = 32;                           // the engine routes a conv here when Cin % HEAD_CH == 0
template <int NPAIR> constexpr size_t head_lds_bytes() { return (size_t)HEAD_HALO_PIX * HeadCfg<NPAIR>::REC; }

template <int NPAIR>   // output pairs: 3 (6 channels) or 2 (3 channels)
__global__ __launch_bounds__(HEAD_NTHR) void head_conv_f32_kernel(const HeadArgs p) {
  extern __shared__ __attribute__((aligned(16))) char hs[];
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int tiles_x = (p.W + TILE_W - 1) / TILE_W, tiles_y = (p.H + HEAD_TH - 1) / HEAD_TH;
  int t = blockIdx.x;
  const int tx_ = t % tiles_x; t /= tiles_x;
  const int ty_ = t % tiles_y;
  const int nb = t / tiles_y;
  const int x0 = tx_ * TILE_W, y0 = ty_ * HEAD_TH;
  const int px = tid & 31, py = tid >> 5;                   // this lane's pixel of the tile
  f2 acc[NPAIR];
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) acc[k] = f2{p.bias[2 * k], p.bias[2 * k + 1]};
  // loader: unit u = tid + 256 * i -> halo pixel u / SLOTS, 16-byte slot u % SLOTS: consecutive lanes fetch the
  // contiguous bytes of one pixel's chunk
  constexpr int HCH = HeadCfg<NPAIR>::CH, HREC = HeadCfg<NPAIR>::REC, SLOTS = HeadCfg<NPAIR>::SLOTS;
  constexpr int NU = (HEAD_HALO_PIX * SLOTS + HEAD_NTHR - 1) / HEAD_NTHR;
  // (80-byte records: the records of a ds_write_b128 phase are spread over their block of 16 as in conv3x3.h)
  auto hp_of = [](int r) { return SLOTS == 4 ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r; };
  int src[NU];                                              // pixel index in the image, -1: padding / nothing
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int hp = hp_of(tid / SLOTS + (HEAD_NTHR / SLOTS) * i);
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    src[i] = (hp < HEAD_HALO_PIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? (nb * p.H + gy) * p.W + gx : -1;
  }
  const int slot = tid % SLOTS;
  f32x4 r[NU];
  auto load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (src[i] >= 0) r[i] = *reinterpret_cast<const f32x4*>(p.in + (size_t)src[i] * p.Cin + c0 + 4 * slot);
    }
  };
  load(0);
  for (int c0 = 0; c0 < p.Cin; c0 += HCH) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int hp = hp_of(tid / SLOTS + (HEAD_NTHR / SLOTS) * i);
      if (hp < HEAD_HALO_PIX) {
        f32x4 v = r[i];
        if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(hs + hp * HREC + slot * 16) = v;
      }
    }
    __syncthreads();
    if (c0 + HCH < p.Cin) load(c0 + HCH);                   // the next chunk's loads fly under this chunk's FMAs
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const char* rec = hs + ((py + tap / 3) * HALO_W + px + tap % 3) * HREC;
      const float* wt = p.w + ((size_t)tap * p.Cin + c0) * (2 * NPAIR);  // uniform: scalar loads
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(rec + q * 16);
        const f2 xlo = {x.x, x.y}, xhi = {x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* we = wt + (4 * q + e) * (2 * NPAIR);
#pragma unroll
          for (int k = 0; k < NPAIR; ++k) {
            const f2 wp = {we[2 * k], we[2 * k + 1]};
            // low half: x_e * w[2k], high half: x_e * w[2k + 1] -- op_sel picks x_e out of its register pair for both
            if ((e & 1) == 0)
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[k]) : "v"(e < 2 ? xlo : xhi), "s"(wp));
            else
              asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[k]) : "v"(e < 2 ? xlo : xhi), "s"(wp));
          }
        }
      }
    }
  }
  const int x = x0 + px, y = y0 + py;
  if (x >= p.W || y >= p.H) return;
  float* ob = p.out + ((size_t)(nb * p.H + y) * p.W + x) * (size_t)p.out_cstride;
#pragma unroll
  for (int n = 0; n < 2 * NPAIR; ++n)
    if (n < p.Cout) {
      float v = (n & 1) ? acc[n >> 1].y : acc[n >> 1].x;
      if (p.relu_out) v = fmaxf(v, 0.f);
      ob[n + p.out_coff + (n >= p.out_split ? p.out_gap : 0)] = v;
    }
}

}  // namespace fisr
