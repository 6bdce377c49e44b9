// DIAGNOSTIC, not part of the product (r04): the fp32 heads fed by LDS-DMA.  Built to lift head_conv.h's register-staged loop
// (3.1 TB/s) towards the HBM roofline; measured SLOWER in the network: 12 head launches of a step 13.3 ms (1105 us average) against
// 10.9 ms (907 us) for the register-staged 8 x 32 kernel, parity green (tests/test_gpu_parity.py, 154 passed).  Why: two 39 KB
// stages per 512-lane workgroup leave room for two workgroups per CU, i.e. one chunk (39 KB) in flight per workgroup and 16 waves per
// CU, where the register-staged kernel has five workgroups per CU (20 waves, 109 KB of halo chunks in flight in registers + LDS);
// the kernel needs ~11 B/clk/CU to keep its FMAs busy and a chunk's DRAM round trip is longer than a chunk's FMAs.  A persistent
// tile loop with the next tile's first chunk prefetched (this version) recovered 4 %; more stages do not fit beside a second
// workgroup.  Also recorded here: behind an asm volatile with a "memory" clobber the compiler no longer proves a global load
// unclobbered and turns uniform weight loads into vector loads (4.5 ms per launch) -- uniform loads then have to go through the
// CONSTANT address space (address_space(4)) to stay scalar.
// To try it again: include this header behind head_conv.h and launch head_conv_dma_kernel<NPAIR, RELU_IN> with
// grid = min(tiles, 2 * CUs), 512 threads, head_dma_lds_bytes() of dynamic LDS, (HeadArgs, n_tiles).
#pragma once
#include "../head_conv.h"

namespace fisr {

// ---- r04: the same arithmetic fed by LDS-DMA ----
// What bounded the kernel above was its staging loop (global -> registers -> ds_write -> barrier -> FMAs -> barrier, 3.1 TB/s with
// everything else ablated), not its FMAs.  Here a 16 x 32 pixel tile (512 lanes, halo 18 x 34: 1.195 x re-read instead of the 8 x 32
// tile's 1.33 x) gets its 16-channel chunks by `buffer_load_dwordx4 ... lds` (39 one-KB copies per chunk, five per wave, no staging
// registers, out-of-image halo pixels point behind the buffer and arrive as zeros), two stages: the copies of chunk k + 1 fly under the
// FMAs of chunk k, ONE barrier per chunk, two workgroups per CU.  An LDS-DMA writes 1 KB contiguously, so the 64-byte pixel records
// cannot be padded; instead the four 16-byte slots of record p are XOR-rotated by (p >> 2) & 3 on the SOURCE side (which channel quad a
// lane fetches), which puts the 16 lanes of every ds_read_b128 phase (16 consecutive pixels, same quad) on 16 distinct 16-byte bank
// groups.  Weights: scalar loads as above.
constexpr int HD_TH = 16, HD_TW = 32, HD_HH = HD_TH + 2, HD_HW = HD_TW + 2, HD_HALO = HD_HH * HD_HW;      // 612 halo pixels
constexpr int HD_CH = 16;                                         // channels per chunk: 64-byte records
constexpr int HD_COPIES = (HD_HALO * 4 + 63) / 64;                // 39 wave copies (2496 units of 16 bytes, 2448 used)
constexpr int HD_STAGE = HD_COPIES * 1024;                        // 39936
constexpr size_t head_dma_lds_bytes() { return 2 * (size_t)HD_STAGE; }

#define FISR_HD_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_HD_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen lds\n\t"
#define FISR_HD_NEXT               "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
#define FISR_HD_END(KEEP)          "s_mov_b32 m0, %[" #KEEP "]"

template <int NPAIR, bool RELU_IN>
__global__ __launch_bounds__(512) void head_conv_dma_kernel(const HeadArgs p, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char hs[];
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  typedef const float __attribute__((address_space(4))) * cptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.W + HD_TW - 1) / HD_TW, tiles_y = (p.H + HD_TH - 1) / HD_TH;
  const size_t img_elems = (size_t)p.H * p.W * p.Cin;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)hs + (unsigned)wave * 1024u;
  constexpr unsigned OOB = 0x80000000u;
  // PERSISTENT: workgroup b walks tiles b, b + grid, ... and the copies of a tile's first chunk fly under the last chunk of the tile
  // before it -- with one workgroup per tile every tile began with an exposed DRAM round trip that two workgroups per CU cannot hide.
  // copies wave, wave + 8, ...: unit u = 64 * copy + lane -> halo pixel u / 4, physical slot u % 4 <- channel quad (u % 4) ^ ((u / 4 >> 2) & 3)
  unsigned hoff[5];
  __amdgpu_buffer_rsrc_t rs;
  auto tile_geom = [&](int t, int& x0, int& y0, int& nb) {
    const int tx_ = t % tiles_x; t /= tiles_x;
    x0 = tx_ * HD_TW; y0 = (t % tiles_y) * HD_TH; nb = t / tiles_y;
  };
  auto set_tile = [&](int t) {
    int x0, y0, nb;
    tile_geom(t, x0, y0, nb);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int u = 64 * (wave + 8 * i) + lane, hp = u >> 2, q = (u & 3) ^ ((hp >> 2) & 3);
      const int hy = hp / HD_HW, hx = hp - hy * HD_HW;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = hp < HD_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      hoff[i] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.Cin + 4u * (unsigned)q) * 4u : OOB;
    }
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)nb * img_elems), 0, (unsigned)(img_elems * 4), 0x00020000);
  };
  auto copy_chunk = [&](int kc, int stage) {
    const unsigned so = (unsigned)kc * (unsigned)(HD_CH * 4), lh = lds0 + (unsigned)stage * (unsigned)HD_STAGE;
    unsigned keep;
    if (wave < 7) {
      asm volatile(FISR_HD_BEGIN(keep, lds) FISR_HD_COPY(o0, rs, so) FISR_HD_NEXT FISR_HD_COPY(o1, rs, so) FISR_HD_NEXT FISR_HD_COPY(o2, rs, so)
                   FISR_HD_NEXT FISR_HD_COPY(o3, rs, so) FISR_HD_NEXT FISR_HD_COPY(o4, rs, so) FISR_HD_END(keep)
                   : [keep] "=&s"(keep) : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]), [o2] "v"(hoff[2]),
                     [o3] "v"(hoff[3]), [o4] "v"(hoff[4]) : "memory", "scc");
    } else {
      asm volatile(FISR_HD_BEGIN(keep, lds) FISR_HD_COPY(o0, rs, so) FISR_HD_NEXT FISR_HD_COPY(o1, rs, so) FISR_HD_NEXT FISR_HD_COPY(o2, rs, so)
                   FISR_HD_NEXT FISR_HD_COPY(o3, rs, so) FISR_HD_END(keep)
                   : [keep] "=&s"(keep) : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]), [o2] "v"(hoff[2]),
                     [o3] "v"(hoff[3]) : "memory", "scc");
    }
  };
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  set_tile(tile);
  copy_chunk(0, 0);
  const int px = tid & 31, py = tid >> 5;                   // this lane's pixel of the tile
  // byte address of quad 0 of the record under each tap; quad q sits at that address ^ (q << 4)
  unsigned ta[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int hp = (py + tap / 3) * HD_HW + px + tap % 3;
    ta[tap] = (unsigned)hp * 64u + (unsigned)((hp >> 2) & 3) * 16u;
  }
  const int nch = p.Cin / HD_CH;
  int stage = 0;
  for (; tile < n_tiles; tile += gridDim.x) {
    int x0, y0, nb;
    tile_geom(tile, x0, y0, nb);
    f2 acc[NPAIR];
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) acc[k] = f2{((cptr_t)(unsigned long long)p.bias)[2 * k], ((cptr_t)(unsigned long long)p.bias)[2 * k + 1]};
    for (int kc = 0; kc < nch; ++kc, stage ^= 1) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's copies of this chunk have landed (and its stores of the tile before)
      __builtin_amdgcn_s_barrier();                                     // ... everyone's, and everyone is done reading the other stage
      asm volatile("" ::: "memory");
      if (kc + 1 < nch) copy_chunk(kc + 1, stage ^ 1);
      else if (tile + (int)gridDim.x < n_tiles) { set_tile(tile + gridDim.x); copy_chunk(0, stage ^ 1); }
      const char* sb = hs + stage * HD_STAGE;
      const int c0 = kc * HD_CH;
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        // uniform -> scalar loads; through the CONSTANT address space: behind the copies' asm ("memory") a plain global load is no
        // longer provably unclobbered and the compiler falls back to one vector load per weight quad (measured: 4.5 ms per launch)
        const cptr_t wt = (cptr_t)(unsigned long long)(p.w + ((size_t)tap * p.Cin + c0) * (2 * NPAIR));
        unsigned a0 = ta[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) a0 = tap == k ? ta[k] : a0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 x = *reinterpret_cast<const f32x4*>(sb + (a0 ^ ((unsigned)q << 4)));
          if (RELU_IN) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
          const f2 xlo = {x.x, x.y}, xhi = {x.z, x.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const cptr_t we = wt + (4 * q + e) * (2 * NPAIR);
#pragma unroll
            for (int k = 0; k < NPAIR; ++k) {
              const f2 wp = {we[2 * k], we[2 * k + 1]};
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
    if (x < p.W && y < p.H) {
      float* ob = p.out + ((size_t)(nb * p.H + y) * p.W + x) * (size_t)p.out_cstride;
#pragma unroll
      for (int n = 0; n < 2 * NPAIR; ++n)
        if (n < p.Cout) {
          float v = (n & 1) ? acc[n >> 1].y : acc[n >> 1].x;
          if (p.relu_out) v = fmaxf(v, 0.f);
          ob[n + p.out_coff + (n >= p.out_split ? p.out_gap : 0)] = v;
        }
    }
  }
}

}  // namespace fisr
