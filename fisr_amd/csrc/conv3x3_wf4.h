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
// ten times F(2x2)'s rounding error per convolution (the whole network: ~2e-5 instead of 1.5e-6 against the fp64 oracle -- what
// the bf16x3 engine has, 7e-5 dB) -- which is why this is its own engine (FISR_PREC_F32W4) and not a silent change of FISR_PREC_F32W.
//
// GEMM view, per transform position p = 0..35:  M_p[co][tile] = sum_ci U_p[co][ci] * V_p[ci][tile].
// Work item = 16 x 32 output pixels (4 x 8 tiles of 4 x 4) x 64 output channels; 512 threads = 8 waves = two per SIMD.
//   * A wave owns ALL 36 positions of a 16-channel x 16-tile block (36 accumulators of v_mfma_f32_16x16x4_f32 = 144 registers):
//     the output transform A^T M A then happens in the wave's own registers -- no exchange between waves, no LDS round trip, no
//     transposition (conv3x3_wino8p.h splits the positions over two waves and spends 10-14 % of a 64-channel item on that
//     exchange and on the DPP transposes behind it).  The accumulator layout of the 16x16 MFMA gives a lane 4 consecutive channels
//     of one tile: every output pixel is one 16-byte store, four lanes cover a 64-byte record.
//   * K loop over 4-channel chunks (= the K of one MFMA), ONE barrier per chunk; LDS (141 312 B):
//       U[2]    36 positions x 64 channels x 4 ci       LDS-DMA of the host-made slab (the LDS image), one chunk ahead
//       V[2]    36 positions x 32 tiles x 4 ci          B^T d B of the NEXT chunk, computed by waves 0-1 while everybody multiplies
//       RAW[3]  18 x 34 halo pixels x 16 B              LDS-DMA from the activation tensor, three chunks ahead
//     Every global byte goes global -> LDS by `buffer_load_dwordx4 ... lds` from inline asm (see conv3x3_dma.h: hidden from the
//     compiler, counted by hand; a lane whose offset lies behind the buffer's end writes ZEROS -- the zero padding is free).
//   * Fragments are 16-byte reads of four consecutive positions (9 + 9 ds_read_b128 feed the 36 MFMAs of a chunk): U as
//     [position quad][channel quarter][k][16 channels][4], V as [position quad][tile half][slot(k, tile)][4] with
//     slot = 16 k + (tile ^ 2 k): the transform lanes (2 tiles x 4 channels per ds_write_b128 phase) and the MFMA lanes (the
//     lane groups of a ds_read_b128 phase: 8 tiles of one k and the 8 other tiles of the next) both hit every bank once.
//   * The raw halo rows hold their columns grouped by column mod 4, so the 8 tiles x 4 channels of a ds_read_b32 phase read 32
//     consecutive dwords.
#pragma once
#include "conv3x3.h"

namespace fisr {

typedef __attribute__((address_space(3))) void* wf4_lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int F4_TH = 16, F4_TW = 32;                  // output pixels of a work item
constexpr int F4_HH = F4_TH + 2, F4_HW = F4_TW + 2;    // halo tile 18 x 34
constexpr int F4_HALO = F4_HH * F4_HW;                 // 612 pixels (16 bytes each per chunk)
constexpr int F4_CH = 4;                               // channels per K chunk
constexpr int F4_BN = 64;                              // output channels per work item
constexpr int F4_RAW_COPIES = 10;                      // 1 KB LDS-DMA copies per raw chunk (640 slots, 612 used)
constexpr int F4_RAW_BYTES = F4_RAW_COPIES * 1024;     // 10240
constexpr int F4_U_BYTES = 36 * F4_BN * F4_CH * 4;     // 36864: one weight slab = 36 copies
constexpr int F4_V_BYTES = 36 * 32 * F4_CH * 4;        // 18432
constexpr size_t wf4_lds_bytes() { return (size_t)2 * F4_U_BYTES + 2 * F4_V_BYTES + 3 * F4_RAW_BYTES; }   // 141312

#define FISR_F4_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_F4_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen lds\n\t"
#define FISR_F4_NEXT               "s_add_u32 m0, m0, 0x1800\n\ts_nop 0\n\t"
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

template <bool RELU_IN, bool HAS_RES>
__global__ __launch_bounds__(512) void conv3x3_wf4_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sU = smem;
  char* const sV = smem + 2 * F4_U_BYTES;
  char* const sR = smem + 2 * F4_U_BYTES + 2 * F4_V_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work item: XCD-aware order as in conv3x3_dma.h (workgroup b runs on XCD b % 8; each XCD walks a contiguous range of
  //      virtual ids in which the N blocks of one pixel tile are neighbours)
  const int tiles_x = (p.W + F4_TW - 1) / F4_TW, tiles_y = (p.H + F4_TH - 1) / F4_TH;
  const int nblocks = p.CoutPad / F4_BN;
  int v = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = v & 7, loc = v >> 3;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int t = v / nblocks;
  const int nblk = v - t * nblocks;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int nb = t / tiles_y;
  const int x0 = tx * F4_TW, y0 = ty * F4_TH;
  const int nch0 = p.C0 / F4_CH, nch = (p.C0 + p.C1) / F4_CH;

  unsigned long long t_start = 0, t_first = 0, t_main = 0, t_real = 0;
  if (p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  // =========================== copy side (waves 2-7; cw = 0..5) ===========================
  // raw copy c (0..9) moves halo slots 64 c .. 64 c + 63; wave cw issues copy cw and (cw < 4) copy cw + 6.
  // slot s -> halo row s / 34, column from the position inside the row: columns 0,4,..,32 | 1,5,..,33 | 2,..,30 | 3,..,31.
  // weight copy c (0..35) is linear; wave cw issues copies cw, cw + 6, ..., cw + 30.
  const int cw = wave - 2;
  constexpr unsigned OOB = 0x80000000u;
  int rpix[2] = {-1, -1};
  if (wave >= 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int s = 64 * (cw + 6 * q) + lane;
      const int py = s / F4_HW, r = s - py * F4_HW;
      const int px = r < 9 ? 4 * r : r < 18 ? 4 * (r - 9) + 1 : r < 26 ? 4 * (r - 18) + 2 : 4 * (r - 26) + 3;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = s < F4_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      rpix[q] = ok ? gy * p.W + gx : -1;
    }
  }
  const size_t img_px = (size_t)p.H * p.W;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.in0 + (size_t)nb * img_px * p.C0), 0,
                                                                       (unsigned)(img_px * p.C0 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in1 ? (const float*)p.in1 + (size_t)nb * img_px * p.C1 : (const float*)p.in0), 0, (unsigned)(img_px * (p.in1 ? p.C1 : p.C0) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (unsigned)((size_t)nch * nblocks * F4_U_BYTES), 0x00020000);
  const unsigned raw_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sR + (unsigned)cw * 1024u;
  const unsigned u_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sU + (unsigned)cw * 1024u;
  const unsigned u_voff = (unsigned)lane * 16u;

  auto copy_raw = [&](int kc, int slot) {
    const bool first = kc < nch0;
    const unsigned so = (unsigned)(first ? kc : kc - nch0) * 16u;
    const unsigned csb = (unsigned)(first ? p.C0 : p.C1) * 4u;
    const unsigned o0 = rpix[0] < 0 ? OOB : (unsigned)rpix[0] * csb;
    const unsigned o1 = rpix[1] < 0 ? OOB : (unsigned)rpix[1] * csb;
    const unsigned lds = raw_lds0 + (unsigned)slot * (unsigned)F4_RAW_BYTES;
    unsigned keep;
#define FISR_F4_RAW(RS)                                                                                                      \
    if (cw < 4) {                                                                                                            \
      asm volatile(FISR_F4_BEGIN(keep, lds) FISR_F4_COPY(o0, rs, so) FISR_F4_NEXT FISR_F4_COPY(o1, rs, so) FISR_F4_END(keep) \
                   : [keep] "=&s"(keep) : [rs] "s"(RS), [so] "s"(so), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1) : "memory", "scc"); \
    } else {                                                                                                                 \
      asm volatile(FISR_F4_BEGIN(keep, lds) FISR_F4_COPY(o0, rs, so) FISR_F4_END(keep)                                       \
                   : [keep] "=&s"(keep) : [rs] "s"(RS), [so] "s"(so), [lds] "s"(lds), [o0] "v"(o0) : "memory", "scc");       \
    }
    if (first) { FISR_F4_RAW(rs0) } else { FISR_F4_RAW(rs1) }
#undef FISR_F4_RAW
  };
  auto copy_u = [&](int kc, int buf) {
    const unsigned s0 = (unsigned)(((size_t)kc * nblocks + nblk) * F4_U_BYTES) + (unsigned)cw * 1024u;
    const unsigned s1 = s0 + 6144u, s2 = s0 + 12288u, s3 = s0 + 18432u, s4 = s0 + 24576u, s5 = s0 + 30720u;
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)F4_U_BYTES;
    unsigned keep;
    asm volatile(FISR_F4_BEGIN(keep, lds) FISR_F4_COPY(o, rs, s0) FISR_F4_NEXT FISR_F4_COPY(o, rs, s1) FISR_F4_NEXT FISR_F4_COPY(o, rs, s2)
                 FISR_F4_NEXT FISR_F4_COPY(o, rs, s3) FISR_F4_NEXT FISR_F4_COPY(o, rs, s4) FISR_F4_NEXT FISR_F4_COPY(o, rs, s5) FISR_F4_END(keep)
                 : [keep] "=&s"(keep)
                 : [rs] "s"(rsw), [lds] "s"(lds), [o] "v"(u_voff), [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2), [s3] "s"(s3), [s4] "s"(s4), [s5] "s"(s5)
                 : "memory", "scc");
  };
  auto wait_keep_youngest_raw = [&]() {          // everything but the raw chunk requested last has landed
    if (cw < 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // =========================== transform side (waves 0-1: tile half th_t = wave) ===========================
  // lane -> (channel of the chunk, tile): 8 tiles of one tile row x 4 channels per ds_read_b32 phase
  const int t_ch = lane & 3, t_t16 = lane >> 2;
  const int t_ty = 2 * (wave & 1) + (t_t16 >> 3), t_tx = t_t16 & 7;
  const int t_roff = ((4 * t_ty) * F4_HW + t_tx) * 16 + t_ch * 4;
  const int t_voff = (wave & 1) * 1024 + (t_ch * 16 + (t_t16 ^ (t_ch << 1))) * 16;
  auto transform = [&](int slot, int vbuf) {
    const char* rb = sR + slot * F4_RAW_BYTES + t_roff;
    float d[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int cb = (j & 3) == 0 ? 0 : (j & 3) == 1 ? 9 : (j & 3) == 2 ? 18 : 26;     // column 4 tx + j sits at cb + tx + j / 4
        d[r][j] = *reinterpret_cast<const float*>(rb + (r * F4_HW + cb + (j >> 2)) * 16);
      }
    if constexpr (RELU_IN) {
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[r][j] = fmaxf(d[r][j], 0.f);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) wf4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
    char* vb = sV + vbuf * F4_V_BYTES + t_voff;
#pragma unroll
    for (int i = 0; i < 6; ++i) wf4_bt(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
#pragma unroll
    for (int q = 0; q < 9; ++q)          // positions 4 q .. 4 q + 3 (position = 6 i + j)
      *reinterpret_cast<f32x4*>(vb + q * 2048) = f32x4{d[(4 * q) / 6][(4 * q) % 6], d[(4 * q + 1) / 6][(4 * q + 1) % 6],
                                                       d[(4 * q + 2) / 6][(4 * q + 2) % 6], d[(4 * q + 3) / 6][(4 * q + 3) % 6]};
  };

  // =========================== MFMA side (all waves) ===========================
  const int cq = wave & 3, th = wave >> 2;        // 16-channel quarter, 16-tile half
  const char* const fu = sU + cq * 1024 + lane * 16;
  const char* const fv = sV + th * 1024 + ((lane & 0x30) | ((lane & 15) ^ ((lane >> 4) << 1))) * 16;
  f32x4 acc[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfma_chunk = [&](int buf) {
    const char* ub = fu + buf * F4_U_BYTES;
    const char* vb = fv + buf * F4_V_BYTES;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ub + q * 4096);
      const f32x4 b = *reinterpret_cast<const f32x4*>(vb + q * 2048);
      acc[4 * q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[4 * q], 0, 0, 0);
      acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[4 * q + 1], 0, 0, 0);
      acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[4 * q + 2], 0, 0, 0);
      acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[4 * q + 3], 0, 0, 0);
    }
  };

  // ---- prologue: raw(0), U(0), raw(1), raw(2) requested; raw(0) -> V[0] ----
  auto clampc = [&](int kc) { return kc < nch ? kc : nch - 1; };     // behind the last chunk the copy COUNT stays fixed (counted waits)
  if (wave >= 2) {
    copy_raw(0, 0);
    copy_u(0, 0);
    copy_raw(clampc(1), 1);
    copy_raw(clampc(2), 2);
    wait_keep_youngest_raw();                      // raw(0), U(0), raw(1) landed
  }
  lds_barrier();
  if (wave < 2) transform(0, 0);
  lds_barrier();
  if (p.trace) t_first = __builtin_readcyclecounter();

  // ---- K loop ----
  int s1 = 1, s2 = 2, s3 = 0;                      // RAW slots of chunks k+1, k+2, k+3
  if (wave < 2) {
    for (int k = 0; k < nch; ++k) {
      if (k + 1 < nch) transform(s1, (k & 1) ^ 1);
      mfma_chunk(k & 1);
      lds_barrier();
      const int s_ = s1; s1 = s2; s2 = s3; s3 = s_;
    }
  } else {
    for (int k = 0; k < nch; ++k) {
      copy_u(clampc(k + 1), (k & 1) ^ 1);
      copy_raw(clampc(k + 3), s3);
      mfma_chunk(k & 1);
      wait_keep_youngest_raw();                    // U(k+1), raw(k+2) landed; raw(k+3) stays in flight
      lds_barrier();
      const int s_ = s1; s1 = s2; s2 = s3; s3 = s_;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (p.trace) t_main = __builtin_readcyclecounter();

  // ---- epilogue: Y = A^T M A in registers, + bias (+ residual), relu, 16-byte stores ----
  // lane: tile th * 16 + (lane & 15) -> tile row / column, channels c0 .. c0 + 3
  const int e_t = th * 16 + (lane & 15);
  const int e_ty = e_t >> 3, e_tx = e_t & 7;
  const int c0 = nblk * F4_BN + cq * 16 + 4 * (lane >> 4);
  const bool c_ok = c0 < p.Cout;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + (c_ok ? c0 : 0));
  const int cq_shift = p.d2s_shift;
  const unsigned sub = (unsigned)c0 >> cq_shift;
  const unsigned sA = p.d2s ? (unsigned)(4 * p.W) << cq_shift : (unsigned)p.W * (unsigned)p.Cout;
  const unsigned sB = p.d2s ? 2u << cq_shift : (unsigned)p.Cout;
  const unsigned vC = p.d2s ? ((((sub >> 1) * 2u * (unsigned)p.W + (sub & 1u)) << cq_shift) + ((unsigned)c0 & ((1u << cq_shift) - 1u))) : (unsigned)c0;
  const unsigned out_bytes = p.d2s ? ((unsigned)(4 * p.H * p.W) << cq_shift) * 4u : (unsigned)(p.H * p.W) * (unsigned)p.Cout * 4u;
  const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
      (char*)p.out + (p.d2s ? ((size_t)nb * 2 * p.H * 2 * p.W << cq_shift) * 4 : (size_t)nb * img_px * p.Cout * 4), 0, out_bytes, 0x00020000);
  const float relu_lo = p.relu_out ? 0.f : -__builtin_huge_valf();
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  unsigned off[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = y0 + 4 * e_ty + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = x0 + 4 * e_tx + j;
      off[i][j] = (c_ok & (oy < p.H) & (ox < p.W)) ? ((unsigned)oy * sA + (unsigned)ox * sB + vC) * 4u : OOB;
    }
  }
  f32x4 res[4][4];
  if constexpr (HAS_RES) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((char*)p.res + (size_t)nb * img_px * p.Cout * 4, 0, out_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) res[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[i][j], 0, 0));
  }
  f32x4 y[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float w[6][4];
#pragma unroll
    for (int a = 0; a < 6; ++a)
      wf4_at(acc[a * 6 + 0][r], acc[a * 6 + 1][r], acc[a * 6 + 2][r], acc[a * 6 + 3][r], acc[a * 6 + 4][r], acc[a * 6 + 5][r],
             w[a][0], w[a][1], w[a][2], w[a][3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float o0, o1, o2, o3;
      wf4_at(w[0][j], w[1][j], w[2][j], w[3][j], w[4][j], w[5][j], o0, o1, o2, o3);
      y[0][j][r] = o0; y[1][j][r] = o1; y[2][j][r] = o2; y[3][j][r] = o3;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 o = y[i][j] + bias;
      if constexpr (HAS_RES) o += res[i][j];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], relu_lo);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), os, off[i][j], 0, 2);   // aux 2: nontemporal
    }
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_main; tr[2] = __builtin_readcyclecounter();
    tr[3] = 0; tr[4] = t_first; tr[5] = t_real; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = 0;
  }
}

#undef FISR_F4_BEGIN
#undef FISR_F4_COPY
#undef FISR_F4_NEXT
#undef FISR_F4_END

}  // namespace fisr
