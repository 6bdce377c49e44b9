// Winograd F(2x2,3x3) fp32 convolution, 8-wave workgroups (two waves per SIMD), ONE work item per workgroup: the
// predecessor of conv3x3_wino8p.h, kept for A/B runs and compiled into -DFISR_DIAG builds only.
//
// Same algorithm, LDS layout, host-made weight slabs and LDS-DMA pipeline as diag/conv3x3_wino4.h; what changes is who
// holds the accumulators.  Measured on MI355X (r02, per-workgroup s_memtime traces + ablation builds): with ONE
// wave per SIMD nothing overlaps with the fp32 MFMAs of that wave -- every s_waitcnt, every VALU instruction of the
// input transform and every copy set-up is added to the 64 x 64 cycles of MFMA per chunk (6.7k instead of 4.1k
// cycles per 8-channel chunk).  Here a workgroup is 512 threads: the 16 transform positions are split between two
// waves of a SIMD (8 accumulators = 128 registers each), so one wave's stalls are covered by the other's MFMAs:
//
//   wave = 4*ph + 2*nh + wh     ph: positions 8*ph .. 8*ph+7 (rows 2*ph, 2*ph+1 of the 4x4 transform grid)
//                               nh: 32 of the 64 output channels     wh: 32 of the 64 Winograd tiles
//   waves 0-3 (ph = 0) compute the input transform of the next chunk between their MFMAs,
//   waves 4-7 (ph = 1) issue the LDS-DMA copies (weight slab of the next chunk, raw halo three chunks ahead),
//                      wait for them and fix the zero padding.
//   (Spreading transform and copies evenly over all 8 waves -- 8-byte LDS accesses, half the work per thread --
//   was measured slower, 6.0k against 5.8k cycles per chunk: on this pipe every non-MFMA instruction of a SIMD costs
//   issue time whoever executes it, and the even split doubles the LDS instruction count.)
//
// The output transform Y = A^T M A needs all four rows of M: s0 = m0 + m1 + m2 (output row 0), s1 = m1 - m2 - m3
// (row 1).  After the K loop the two waves of a pair swap one row each through LDS (16 KB per wave, the V/U buffers
// are free by then): the ph = 0 wave receives m2 and finishes output row 0, the ph = 1 wave receives m1 and
// finishes row 1.
#pragma once
#include "../conv3x3_wino_common.h"

namespace fisr {

template <bool RELU_IN>
__global__ __launch_bounds__(512, 2) void conv3x3_wino8_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sV = smem;
  char* const sU = smem + 2 * W_SLAB;
  char* const sR = smem + 4 * W_SLAB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int kh = lane >> 5;
  const int wh = wave & 1;          // which 32 wtiles (pixel rows 0-3 / 4-7 of the tile)
  const int nh = (wave >> 1) & 1;   // which 32 of the 64 output channels
  const int ph = wave >> 2;         // which 8 of the 16 transform positions

  const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
  const int tiles_y = (p.H + TILE_H - 1) / TILE_H;
  int v = blockIdx.x;               // XCD-aware work order, as in conv3x3.h (speed only)
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = v & 7, loc = v >> 3;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int nblocks = p.CoutPad / W_BN;
  int t = v / nblocks;
  const int nblk = v - t * nblocks;
  const int n0 = nblk * W_BN;
  const int tx_ = t % tiles_x; t /= tiles_x;
  const int ty_ = t % tiles_y;
  const int nb = t / tiles_y;
  const int x0 = tx_ * TILE_W, y0 = ty_ * TILE_H;

  unsigned long long t_start = 0, t_main = 0, t_first = 0, t_real = 0;
  if (p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  const int nch = (p.C0 + p.C1) / W_CH;

  // =========================== copy side (waves 4-7; ct = thread index among them) ===========================
  // raw halo chunk: 680 16-byte units, unit u = ct + 256*i -> halo pixel u >> 1, half u & 1.  Clamped source
  // addresses (fixed number of copies per wave), zeros written over padding units once the copy has landed.
  const int ct = tid & 255;
  const int cw = wave & 3;                              // copy wave 0..3: cw < 2 three copies, 2: 2 + 40 lanes, 3: two
  unsigned raw_voff0[3], raw_voff1[3];
  bool in_ok[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = ct + 256 * i;
    const int pix = min(u >> 1, HALO_PIX - 1);
    const int py = pix / HALO_W, px = pix - py * HALO_W;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    in_ok[i] = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const unsigned gp = (unsigned)((nb * p.H + min(max(gy, 0), p.H - 1)) * p.W + min(max(gx, 0), p.W - 1));
    raw_voff0[i] = (gp * (unsigned)p.C0 + (unsigned)(ct & 1) * 4u) * 4u;
    raw_voff1[i] = (gp * (unsigned)p.C1 + (unsigned)(ct & 1) * 4u) * 4u;
  }
  const unsigned raw_lds0 = (unsigned)(size_t)(lds_ptr_t)sR + (unsigned)cw * 1024u;
  auto copy_raw = [&](int kc, int slot) {
    const int c0 = min(kc, nch - 1) * W_CH;          // past the last chunk: a harmless repeat of the last one
    const bool first = c0 < p.C0;
    const char* g = first ? (const char*)p.in0 + (size_t)c0 * 4 : (const char*)p.in1 + (size_t)(c0 - p.C0) * 4;
    const unsigned o0 = first ? raw_voff0[0] : raw_voff1[0];
    const unsigned o1 = first ? raw_voff0[1] : raw_voff1[1];
    const unsigned o2 = first ? raw_voff0[2] : raw_voff1[2];
    const unsigned lds = raw_lds0 + (unsigned)slot * (unsigned)W_RAW;
    unsigned keep;
    if (cw < 2) {
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g)
                   FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o2, g) FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2) : "memory", "scc");
    } else if (cw == 2) {
      unsigned long long ex;
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g)
                   FISR_GLDS_NEXT_ROW
                   "s_mov_b64 %[ex], exec\n\ts_bfm_b64 exec, 40, 0\n\t"      // units 640..679: lanes 0..39
                   FISR_GLDS_COPY(o2, g)
                   "s_mov_b64 exec, %[ex]\n\t" FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep), [ex] "=&s"(ex) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2)
                   : "memory", "scc");
    } else {
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g) FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1) : "memory", "scc");
    }
  };
  const bool any_pad = !(in_ok[0] && in_ok[1] && (in_ok[2] || ct + 512 >= W_RAW_UNITS));
  auto fix_raw = [&](int slot) {
    if (any_pad) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int u = ct + 256 * i;
        if (!in_ok[i] && u < W_RAW_UNITS) *reinterpret_cast<f32x4*>(sR + slot * W_RAW + u * 16) = z;
      }
    }
  };
  // U slab: 2048 16-byte units, a linear copy of the host-made LDS image of (chunk kc, N-block)
  const char* const u_base = (const char*)p.wpk + (size_t)nblk * W_SLAB;
  const size_t u_stride = (size_t)nblocks * W_SLAB;
  const unsigned u_lds0 = (unsigned)(size_t)(lds_ptr_t)sU + (unsigned)cw * 1024u;
  const unsigned u_voff = (unsigned)ct * 16u;
  auto copy_u = [&](int kc, int buf) {
    const char* g = u_base + (size_t)kc * u_stride;
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)W_SLAB;
    unsigned keep;
    asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g) FISR_GLDS_NEXT_ROW
                 FISR_GLDS_COPY(o2, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o3, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o4, g)
                 FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o5, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o6, g) FISR_GLDS_NEXT_ROW
                 FISR_GLDS_COPY(o7, g) FISR_GLDS_END(keep)
                 : [keep] "=&s"(keep)
                 : [g] "s"(g), [lds] "s"(lds), [o0] "v"(u_voff), [o1] "v"(u_voff + 0x1000u), [o2] "v"(u_voff + 0x2000u),
                   [o3] "v"(u_voff + 0x3000u), [o4] "v"(u_voff + 0x4000u), [o5] "v"(u_voff + 0x5000u),
                   [o6] "v"(u_voff + 0x6000u), [o7] "v"(u_voff + 0x7000u)
                 : "memory", "scc");
  };
  auto wait_copies_keep_youngest_raw = [&]() {       // all copies but the youngest raw chunk's have landed
    if (cw < 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // =========================== transform side (waves 0-3: thread = (wtile, channel quad, row half)) ==========
  // B^T rows: T0 = d0 - d2, T1 = d1 + d2, T2 = d2 - d1, T3 = d1 - d3 (then the same along columns).  Row half 0
  // (waves 0, 1) makes T0, T1 of its wtile, row half 1 (waves 2, 3) T2, T3; per wave the row roles are uniform:
  //   A = d[ra_x] - d[ra_z]        B = d[1] + sgn * d[rb_z]
  const int t_cq = tid & 1, t_w = (tid >> 1) & 63;
  const int t_rh = __builtin_amdgcn_readfirstlane((tid >> 7) & 1);
  const int t_ty = t_w >> 4, t_tx = t_w & 15;
  const int ra_x = t_rh ? 2 : 0, ra_z = t_rh ? 1 : 2, rb_z = t_rh ? 3 : 2;
  const float sgn = t_rh ? -1.f : 1.f;
  const int t_roff = ((2 * t_ty) * HALO_W + 2 * t_tx) * W_REC + t_cq * 16;
  const int t_voff = ((8 * t_rh) * 64 + t_w) * W_REC + ((t_cq ^ ((t_w >> 3) & 1)) * 16);
  // One v_max_f32 per value (fmaxf() costs two: it canonicalises its operand first) and packed fp32 adds
  // (v_pk_add_f32: two values per instruction): on this pipe every VALU instruction of a SIMD is time the fp32
  // MFMAs do not get (they run on the same lanes), so the transform is written for instruction count.
  auto relu4 = [&](f32x4 f) {
    if constexpr (RELU_IN) {
      asm("v_max_f32 %0, 0, %0" : "+v"(f.x)); asm("v_max_f32 %0, 0, %0" : "+v"(f.y));
      asm("v_max_f32 %0, 0, %0" : "+v"(f.z)); asm("v_max_f32 %0, 0, %0" : "+v"(f.w));
    }
    return f;
  };
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  auto sub4 = [&](f32x4 a, f32x4 b) {
    f32x2_ lo, hi;
    const f32x2_ alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
  };
  const f32x2_ sgn2 = {sgn, sgn};
  auto fma4_sgn = [&](f32x4 z, f32x4 y) {          // y + sgn * z
    f32x2_ lo, hi;
    const f32x2_ zlo = {z.x, z.y}, zhi = {z.z, z.w}, ylo = {y.x, y.y}, yhi = {y.z, y.w};
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(zlo), "v"(sgn2), "v"(ylo));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(zhi), "v"(sgn2), "v"(yhi));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
  };
  auto add4 = [&](f32x4 a, f32x4 b) {
    f32x2_ lo, hi;
    const f32x2_ alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(ahi), "v"(bhi));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
  };
  f32x4 txa[2], tza[2], tyb[2], tzb[2], TA[4], TB[4];
  auto tr_read = [&](int slot, int cpair) {
    const char* rb = sR + slot * W_RAW + t_roff;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = 2 * cpair + k;
      txa[k] = *reinterpret_cast<const f32x4*>(rb + (ra_x * HALO_W + c) * W_REC);
      tza[k] = *reinterpret_cast<const f32x4*>(rb + (ra_z * HALO_W + c) * W_REC);
      tyb[k] = *reinterpret_cast<const f32x4*>(rb + (1 * HALO_W + c) * W_REC);
      tzb[k] = *reinterpret_cast<const f32x4*>(rb + (rb_z * HALO_W + c) * W_REC);
    }
  };
  auto tr_rows = [&](int cpair) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      TA[2 * cpair + k] = sub4(relu4(txa[k]), relu4(tza[k]));
      TB[2 * cpair + k] = fma4_sgn(relu4(tzb[k]), relu4(tyb[k]));
    }
  };
  auto tr_cols = [&](int vbuf, int which) {
    char* vw = sV + vbuf * W_SLAB + t_voff + which * 4 * 64 * W_REC;
    const f32x4* T = which ? TB : TA;
    *reinterpret_cast<f32x4*>(vw + 0 * 64 * W_REC) = sub4(T[0], T[2]);
    *reinterpret_cast<f32x4*>(vw + 1 * 64 * W_REC) = add4(T[1], T[2]);
    *reinterpret_cast<f32x4*>(vw + 2 * 64 * W_REC) = sub4(T[2], T[1]);
    *reinterpret_cast<f32x4*>(vw + 3 * 64 * W_REC) = sub4(T[1], T[3]);
  };

  // =========================== MFMA side (all waves) ==========================================================
  // lane (li, kh): A operand (rows) = U record of channel row 32*nh + li, B operand (cols) = V record of wtile
  // 32*wh + li, 16-byte half kh (swizzled by bit 3 of the record index); this wave's positions 8*ph + q, q = 0..7.
  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int f_off = li * W_REC + ((kh ^ ((li >> 3) & 1)) * 16) + (8 * ph) * 64 * W_REC;
  const int fu_off = (32 * nh) * W_REC + f_off;
  const int fv_off = (32 * wh) * W_REC + f_off;
  f32x4 fa[2][2], fb[2][2];                     // [register buffer][position of the pair]
  auto frag_load = [&](int buf, int pp, int rb_) {
    const char* ub = sU + buf * W_SLAB + fu_off + (2 * pp) * 64 * W_REC;
    const char* vb = sV + buf * W_SLAB + fv_off + (2 * pp) * 64 * W_REC;
    fa[rb_][0] = *reinterpret_cast<const f32x4*>(ub);
    fb[rb_][0] = *reinterpret_cast<const f32x4*>(vb);
    fa[rb_][1] = *reinterpret_cast<const f32x4*>(ub + 64 * W_REC);
    fb[rb_][1] = *reinterpret_cast<const f32x4*>(vb + 64 * W_REC);
  };
#define FISR_W8_MMA(PP, RB, E)                                                                                     \
  acc[2 * (PP)]     = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[RB][0].E, fb[RB][0].E, acc[2 * (PP)], 0, 0, 0);      \
  acc[2 * (PP) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[RB][1].E, fb[RB][1].E, acc[2 * (PP) + 1], 0, 0, 0);
#define FISR_W8_STAGE(PP) FISR_W8_MMA(PP, (PP) & 1, x) FISR_W8_MMA(PP, (PP) & 1, y) FISR_W8_MMA(PP, (PP) & 1, z) FISR_W8_MMA(PP, (PP) & 1, w)
#define FISR_W8_INTERLEAVE(NOTHER)                                                    \
  _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
    __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200, NOTHER, 0);           \
  }

  // ---- prologue: U(0) and raw(0..2) requested by the copy waves; chunk 0 transformed by the transform waves ----
  // raw chunk c lives in RAW[c % 3]: requested at iteration c-3, landed + padding fixed at the end of iteration
  // c-2, transformed during iteration c-1 (into V[c & 1]), multiplied in iteration c.
  if (ph == 1) {
    copy_raw(0, 0);
    copy_u(0, 0);
    copy_raw(1, 1);
    copy_raw(2, 2);
    if (cw < 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // raw(0), U(0) landed; raw(1), raw(2) in flight
    else        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    fix_raw(0);
  }
  lds_barrier();
  if (ph == 0) {
    tr_read(0, 0); tr_rows(0); tr_read(0, 1); tr_rows(1); tr_cols(0, 0); tr_cols(0, 1);     // chunk 0 -> V[0]
  } else {
    wait_copies_keep_youngest_raw();                                // raw(1) landed; raw(2) in flight
    fix_raw(1);
  }
  lds_barrier();
  if (p.trace) t_first = __builtin_readcyclecounter();

  // ---- main loop: MFMAs of chunk kc || input transform of chunk kc+1 || copies of U(kc+1), raw(kc+3) ----
  int slot1 = 1, slot2 = 2, slot3 = 0;                 // RAW slots of chunks kc+1, kc+2, kc+3
  if (ph == 0) {
    for (int kc = 0; kc + 1 < nch; ++kc) {
      const int b = kc & 1;
      frag_load(b, 0, 0);
      if (!(FISR_WABL & 1)) tr_read(slot1, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        if (pp < 3) frag_load(b, pp + 1, (pp + 1) & 1);
        if (!(FISR_WABL & 1)) {
          if (pp == 0) { tr_rows(0); tr_read(slot1, 1); }
          if (pp == 1) tr_rows(1);
          if (pp == 2) tr_cols(b ^ 1, 0);
          if (pp == 3) tr_cols(b ^ 1, 1);
        }
        if (!(FISR_WABL & 4)) { FISR_W8_STAGE(pp) }
        FISR_W8_INTERLEAVE(4)
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
      const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
    }
  } else {
    for (int kc = 0; kc + 1 < nch; ++kc) {
      const int b = kc & 1;
      frag_load(b, 0, 0);
      if (!(FISR_WABL & 2)) {
        copy_u(kc + 1, b ^ 1);
        copy_raw(kc + 3, slot3);
      }
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        if (pp < 3) frag_load(b, pp + 1, (pp + 1) & 1);
        if (!(FISR_WABL & 4)) { FISR_W8_STAGE(pp) }
      }
      if (!(FISR_WABL & 2)) wait_copies_keep_youngest_raw();      // U(kc+1) and raw(kc+2) landed; raw(kc+3) stays in flight
      fix_raw(slot2);
      lds_barrier();
      const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
    }
  }

  // epilogue geometry: lane (li, kh) of wave (ph, nh, wh) owns wtile 32*wh + li, output row i = ph of its 2x2
  // pixels, i.e. pixels (y0 + 2*ty + ph, x0 + 2*tx + j), j = 0, 1, and the 16-channel record c0 .. c0+15.
  // Residual records and bias are requested before the last chunk's MFMAs (see conv3x3_wino.h).
  const int w_ = 32 * wh + li;
  const int ty = w_ >> 4, tx = w_ & 15;
  const int c0 = n0 + 32 * nh + 16 * kh;
  const bool c_ok = c0 < p.Cout;
  const int oy = y0 + 2 * ty + ph;
  // Residual records: fetched quad-transposed -- the four lanes of a quad read the four 16-byte units of ONE record
  // per instruction (64 contiguous bytes) -- and transposed back in registers right before use.
  uint4 rres[2][4];
  float bv[16];
  const int txq = tx & ~3;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = x0 + 2 * (txq + k) + j;
      rres[j][k] = make_uint4(0u, 0u, 0u, 0u);
      if (p.res != nullptr && c_ok && oy < p.H && x < p.W)
        rres[j][k] = *(reinterpret_cast<const uint4*>((const float*)p.res + ((size_t)(nb * p.H + oy) * p.W + x) * p.Cout + c0) + (lane & 3));
    }
  {
    const f32x4* bq = reinterpret_cast<const f32x4*>(p.bias + (c_ok ? c0 : 0));
#pragma unroll
    for (int k = 0; k < 4; ++k) { const f32x4 f = bq[k]; bv[4 * k] = f.x; bv[4 * k + 1] = f.y; bv[4 * k + 2] = f.z; bv[4 * k + 3] = f.w; }
  }
  {
    const int b = (nch - 1) & 1;
    frag_load(b, 0, 0);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      if (pp < 3) frag_load(b, pp + 1, (pp + 1) & 1);
      FISR_W8_STAGE(pp)
    }
  }
#undef FISR_W8_MMA
#undef FISR_W8_STAGE
#undef FISR_W8_INTERLEAVE
  if (p.trace) t_main = __builtin_readcyclecounter();

  // ---- row exchange between the two waves of a pair, through the (now free) V/U region ----
  // local rows: acc[0..3] = M row 2*ph (columns 0..3), acc[4..7] = M row 2*ph + 1.
  // ph = 0 sends m1 = acc[4..7] and receives m2;  ph = 1 sends m2 = acc[0..3] and receives m1.
  // slot of wave w: smem + w * 16 KB, unit (c*4 + rq) * 1 KB + lane * 16 = accumulator c, registers 4*rq .. 4*rq+3.
  lds_barrier();                                   // every wave is done reading V / U
  if (!(FISR_WABL & 16)) {
    char* mine = smem + wave * 16384 + lane * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 f;
        if (ph == 0) { f.x = acc[4 + c][4 * rq]; f.y = acc[4 + c][4 * rq + 1]; f.z = acc[4 + c][4 * rq + 2]; f.w = acc[4 + c][4 * rq + 3]; }
        else         { f.x = acc[c][4 * rq];     f.y = acc[c][4 * rq + 1];     f.z = acc[c][4 * rq + 2];     f.w = acc[c][4 * rq + 3]; }
        *reinterpret_cast<f32x4*>(mine + (c * 4 + rq) * 1024) = f;
      }
  }
  lds_barrier();
  {
    const char* theirs = smem + (wave ^ 4) * 16384 + lane * 16;
    const float relu_floor = p.relu_out ? 0.f : -__builtin_huge_valf();
    const int cq_shift = p.d2s_shift;
    auto record = [&](int y, int xc) -> size_t {
      if (p.d2s) {
        const int sub = c0 >> cq_shift, c = c0 & ((1 << cq_shift) - 1);
        return (((size_t)(nb * 2 * p.H + 2 * y + (sub >> 1))) * (2 * p.W) + 2 * xc + (sub & 1)) * ((size_t)1 << cq_shift) + c;
      }
      return ((size_t)(nb * p.H + y) * p.W + xc) * p.Cout + c0;
    };
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    // The lane's two pixels x 64-byte records are first collected (8 x 16 bytes), then transposed inside each quad of
    // lanes so that one store instruction makes the four lanes of a quad write ONE record (64 contiguous bytes)
    // instead of four 16-byte pieces of four records.
    uint4 rec[2][4];
    quad_transpose(rres[0], lane);
    quad_transpose(rres[1], lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {                    // channels 4k .. 4k+3 of the record
      f32x4 got[4];                                  // the partner's row, columns 0..3, these four channels
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (FISR_WABL & 16) got[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        else got[c] = *reinterpret_cast<const f32x4*>(theirs + (c * 4 + k) * 1024);
      }
      f32x4 o0, o1;
      const f32x4 r0 = __builtin_bit_cast(f32x4, rres[0][k]), r1 = __builtin_bit_cast(f32x4, rres[1][k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * k + e;
        float sc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)    // s0 = (m0 + m1) + m2   |   s1 = (m1 - m2) - m3
          sc[c] = ph == 0 ? (acc[c][r] + acc[4 + c][r]) + got[c][e]
                          : (got[c][e] - acc[c][r]) - acc[4 + c][r];
        const float y0v = (sc[0] + sc[1]) + sc[2];
        const float y1v = (sc[1] - sc[2]) - sc[3];
        o0[e] = fmaxf((y0v + bv[r]) + r0[e], relu_floor);
        o1[e] = fmaxf((y1v + bv[r]) + r1[e], relu_floor);
      }
      rec[0][k] = __builtin_bit_cast(uint4, o0);
      rec[1][k] = __builtin_bit_cast(uint4, o1);
    }
    if (FISR_WABL & 8) {            // ablation: no stores (keep the values alive)
      if (rec[0][0].x == 0x12345678u && rec[1][3].y == 0x9abcdef0u) *((uint4*)p.out) = rec[0][1];
    } else if (FISR_WABL & 64) {    // ablation: untransposed 16-byte pieces
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (c_ok && oy < p.H && x0 + 2 * tx + j < p.W) {
          u32x4_t* dst = reinterpret_cast<u32x4_t*>((float*)p.out + record(oy, x0 + 2 * tx + j));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (FISR_WABL & 32) dst[k] = __builtin_bit_cast(u32x4_t, rec[j][k]);
            else __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, rec[j][k]), dst + k);
          }
        }
    } else {
      // after the transpose unit k of this lane belongs to the record of quad lane k: wtile column (tx & ~3) + k
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        quad_transpose(rec[j], lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = x0 + 2 * (txq + k) + j;
          if (c_ok && oy < p.H && x < p.W) {
            u32x4_t* dst = reinterpret_cast<u32x4_t*>((float*)p.out + record(oy, x)) + (lane & 3);
            if (FISR_WABL & 32) *dst = __builtin_bit_cast(u32x4_t, rec[j][k]);
            else __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, rec[j][k]), dst);
          }
        }
      }
    }
  }
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_main; tr[2] = __builtin_readcyclecounter();
    tr[3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    tr[4] = t_first; tr[5] = t_real; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = 0;
  }
}

}  // namespace fisr
