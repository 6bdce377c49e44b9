// Winograd F(2x2,3x3) fp32 convolution, 8-wave workgroups, PERSISTENT: one workgroup per CU walks a list of
// (pixel tile, 64-channel block) work items and never lets its copy / transform pipeline drain.
//
// The one-item-per-workgroup kernel (diag/conv3x3_wino8.h) spends 7.3k cycles of a 60k-cycle 64->64 tile in its prologue (the first HBM round trip and the
// first input transform) and cannot hide it: one 64 wtile x 64 channel x 16 position accumulator set is half the CU's
// register file, so no second workgroup fits beside it.  Here the three streams of the K loop simply continue into
// the NEXT work item of the same workgroup:
//     raw halo chunk g+3   (LDS-DMA, three chunks ahead)        g = global chunk counter over the workgroup's items
//     weight slab    g+1   (LDS-DMA, one ahead)
//     input transform g+1  (waves 0-3, into V[(g+1) & 1])
// so the last three iterations of an item already copy / transform the first chunks of the next one, and only the very
// first item of a workgroup pays a prologue.  The epilogue of an item (output transform + stores) runs while those
// copies are in flight; the row exchange between the two waves of a pair uses the V/U buffers the last chunk has just
// released (64 KB: the column stage of the output transform is done BEFORE the exchange, which halves it).
// LDS layout, host-made weight slabs and the LDS-DMA macros: conv3x3_wino_common.h.  Needs Cin >= 32 (at least 4 chunks,
// so the copy stream never runs more than one work item ahead); layers below that run on the direct kernel (conv3x3.h).
#pragma once
#include "conv3x3_wino_common.h"

namespace fisr {

struct first_t { static constexpr bool value = true; };    // tags: first K iteration of a work item / the others
struct rest_t { static constexpr bool value = false; };

// GENERAL = false: FISRnet (dense NHWC tensors, relu).  GENERAL = true: PWC-Net's layers as well -- channel ranges of a
// wider buffer as input and output (in0_cs, rec_cs, rec_co), leaky relu (slope), dilation (dil).  A template flag because
// the K loops have no register to spare: the extra kernel arguments alone make the compiler spill.
// HAS_RES = false: the layer has no residual input (p.res == NULL) -- nothing is loaded, transposed or added for it: a
// third of the output stage's instructions.  (A kernel template parameter: the same epilogue twice inside one kernel
// makes the compiler spill.)
template <bool RELU_IN, bool GENERAL, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void conv3x3_wino8p_kernel(const ConvArgs p, const int n_items) {
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

  // Dilation d (PWC-Net's context network): the image is d x d interleaved sub-images (pixel (y, x) belongs to sub-image
  // (y % d, x % d)) and a dilated 3x3 convolution is an ordinary one inside each of them -- zero padding included.  The
  // tiles and all tile coordinates below live in a sub-image; only the addresses are scaled back.  d = 1: one sub-image.
  const int dil = GENERAL ? p.dil : 1;                 // (a compile-time 1 for FISRnet: no index arithmetic is added)
  const int in0_cs = GENERAL ? p.in0_cs : p.C0, in1_cs = GENERAL ? p.in1_cs : p.C1;
  const int rec_cs = GENERAL ? p.rec_cs : p.Cout, rec_co = GENERAL ? p.rec_co : 0;
  const float slope = GENERAL ? p.slope : 0.f;
  const int tiles_x = ((p.W + dil - 1) / dil + TILE_W - 1) / TILE_W;
  const int tiles_y = ((p.H + dil - 1) / dil + TILE_H - 1) / TILE_H;
  const int nblocks = p.CoutPad / W_BN;
  const int nch = (p.C0 + p.C1) / W_CH;

  // work item b (0 .. n_items-1) -> (x0, y0, nb, nblk): XCD-aware order as in conv3x3.h -- workgroup w runs on XCD
  // w % 8, gridDim.x is a multiple of 8, so all items of a workgroup stay on one XCD and each XCD walks a contiguous
  // range of virtual ids in which the N-blocks of one pixel tile are consecutive.
  struct Item { int x0, y0, nb, nblk, ry, rx; };       // (x0, y0): in sub-image (ry, rx)
  auto item_of = [&](int b) {
    const int q = n_items >> 3, r = n_items & 7;
    const int xcd = b & 7, loc = b >> 3;
    int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int t = v / nblocks;
    Item it;
    it.nblk = v - t * nblocks;
    const int tx_ = t % tiles_x; t /= tiles_x;
    const int ty_ = t % tiles_y; t /= tiles_y;
    const int sub = t % (dil * dil);
    it.nb = t / (dil * dil);
    it.ry = sub / dil; it.rx = sub - it.ry * dil;
    it.x0 = tx_ * TILE_W; it.y0 = ty_ * TILE_H;
    return it;
  };
  // the ids of this workgroup: blockIdx.x, + gridDim.x, ... while < n_items
  auto valid = [&](int b) { return b < n_items; };

  unsigned long long t_start = 0, t_main = 0, t_first = 0, t_real = 0;
  if (p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  // =========================== copy side (waves 4-7; ct = thread index among them) ===========================
  const int ct = tid & 255;
  const int cw = wave & 3;                              // copy wave 0..3: cw < 2 three copies, 2: 2 + 40 lanes, 3: two
  unsigned raw_gp[3];                                   // clamped pixel index per copy unit, of the item the raw copy stream is in
  bool fix_ok[3];                                       // padding masks of the item whose chunk is being fixed
  bool fix_any = false;
  const char* u_base = nullptr;                         // of the item the weight stream is in
  auto raw_geom = [&](const Item& it) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int u = ct + 256 * i;
      asm volatile("" : "+v"(u));               // recompute per call: hoisted, the halo coordinates would be spilled
      const int pix = min(u >> 1, HALO_PIX - 1);
      const int py = pix / HALO_W, ix = pix - py * HALO_W;
      const int px = ix < HALO_W / 2 ? 2 * ix : 2 * (ix - HALO_W / 2) + 1;     // even columns first, then the odd ones
      const int gy = (it.y0 - 1 + py) * dil + it.ry, gx = (it.x0 - 1 + px) * dil + it.rx;
      raw_gp[i] = (unsigned)((it.nb * p.H + min(max(gy, 0), p.H - 1)) * p.W + min(max(gx, 0), p.W - 1));
    }
  };
  auto fix_geom = [&](const Item& it) {
    fix_any = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int u = ct + 256 * i;
      asm volatile("" : "+v"(u));
      const int pix = min(u >> 1, HALO_PIX - 1);
      const int py = pix / HALO_W, ix = pix - py * HALO_W;
      const int px = ix < HALO_W / 2 ? 2 * ix : 2 * (ix - HALO_W / 2) + 1;     // even columns first, then the odd ones
      const int gy = (it.y0 - 1 + py) * dil + it.ry, gx = (it.x0 - 1 + px) * dil + it.rx;
      fix_ok[i] = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) || u >= W_RAW_UNITS;
      fix_any = fix_any || !fix_ok[i];
    }
  };
  const unsigned raw_lds0 = (unsigned)(size_t)(lds_ptr_t)sR + (unsigned)cw * 1024u;
  auto copy_raw = [&](int kc, int slot) {              // chunk kc (0 .. nch-1) of the item raw_voff* describe
    const int c0 = kc * W_CH;
    const bool first = c0 < p.C0;
    const char* g = first ? (const char*)p.in0 + (size_t)c0 * 4 : (const char*)p.in1 + (size_t)(c0 - p.C0) * 4;
    const unsigned cs = (unsigned)(first ? in0_cs : in1_cs) * 4u, ho = (unsigned)(ct & 1) * 16u;
    const unsigned o0 = raw_gp[0] * cs + ho, o1 = raw_gp[1] * cs + ho, o2 = raw_gp[2] * cs + ho;
    const unsigned lds = raw_lds0 + (unsigned)slot * (unsigned)W_RAW;
    unsigned keep;
    if (cw < 2) {
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY_RAW(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY_RAW(o1, g)
                   FISR_GLDS_NEXT_ROW FISR_GLDS_COPY_RAW(o2, g) FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2) : "memory", "scc");
    } else if (cw == 2) {
      unsigned long long ex;
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY_RAW(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY_RAW(o1, g)
                   FISR_GLDS_NEXT_ROW
                   "s_mov_b64 %[ex], exec\n\ts_bfm_b64 exec, 40, 0\n\t"      // units 640..679: lanes 0..39
                   FISR_GLDS_COPY_RAW(o2, g)
                   "s_mov_b64 exec, %[ex]\n\t" FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep), [ex] "=&s"(ex) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2)
                   : "memory", "scc");
    } else {
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY_RAW(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY_RAW(o1, g) FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1) : "memory", "scc");
    }
  };
  // A landed raw chunk is finished in place by the threads that requested it: relu (RELU_IN: once per element here,
  // instead of once per tile that reads it -- four times -- in the transform waves, whose VALU time is the K loop's
  // critical path), zeros for the padding pixels.  In the K loop the relu is split into a read and a write half with a
  // stage of MFMAs in between (a wave issues in order: waiting for the ds_read would stall its MFMAs).
  f32x4 rl[3];
  auto relu_read = [&](int slot) {
    if constexpr (RELU_IN) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (ct + 256 * i < W_RAW_UNITS) rl[i] = *reinterpret_cast<const f32x4*>(sR + slot * W_RAW + (ct + 256 * i) * 16);
    }
  };
  auto relu_write = [&](int slot) {
    if constexpr (RELU_IN) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (ct + 256 * i < W_RAW_UNITS) {
          f32x4 f = rl[i];
          asm("v_max_f32 %0, 0, %0" : "+v"(f.x)); asm("v_max_f32 %0, 0, %0" : "+v"(f.y));
          asm("v_max_f32 %0, 0, %0" : "+v"(f.z)); asm("v_max_f32 %0, 0, %0" : "+v"(f.w));
          *reinterpret_cast<f32x4*>(sR + slot * W_RAW + (ct + 256 * i) * 16) = f;
        }
    }
  };
  auto zero_padding = [&](int slot) {
    if (fix_any) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (!fix_ok[i]) *reinterpret_cast<f32x4*>(sR + slot * W_RAW + (ct + 256 * i) * 16) = z;
    }
  };
  auto fix_raw = [&](int slot) { relu_read(slot); relu_write(slot); zero_padding(slot); };
  const size_t u_stride = (size_t)nblocks * W_SLAB;
  const unsigned u_lds0 = (unsigned)(size_t)(lds_ptr_t)sU + (unsigned)cw * 1024u;
  const unsigned u_voff = (unsigned)ct * 16u;
  auto copy_u = [&](int kc, int buf) {                 // slab kc of the item u_base describes
    // one VGPR offset, eight scalar row bases (eight VGPR offsets would stay live across the whole loop)
    const char* g = u_base + (size_t)kc * u_stride;
    const char *g1 = g + 0x1000, *g2 = g + 0x2000, *g3 = g + 0x3000, *g4 = g + 0x4000, *g5 = g + 0x5000,
               *g6 = g + 0x6000, *g7 = g + 0x7000;
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)W_SLAB;
    unsigned keep;
    asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o, g1) FISR_GLDS_NEXT_ROW
                 FISR_GLDS_COPY(o, g2) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o, g3) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o, g4)
                 FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o, g5) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o, g6) FISR_GLDS_NEXT_ROW
                 FISR_GLDS_COPY(o, g7) FISR_GLDS_END(keep)
                 : [keep] "=&s"(keep)
                 : [g] "s"(g), [g1] "s"(g1), [g2] "s"(g2), [g3] "s"(g3), [g4] "s"(g4), [g5] "s"(g5), [g6] "s"(g6),
                   [g7] "s"(g7), [lds] "s"(lds), [o] "v"(u_voff)
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

  // =========================== transform side (waves 0-3) — as in conv3x3_wino8.h ===========================
  const int t_cq = tid & 1, t_w = (tid >> 1) & 63;
  const int t_rh = __builtin_amdgcn_readfirstlane((tid >> 7) & 1);
  const int t_ty = t_w >> 4, t_tx = t_w & 15;
  const int ra_x = t_rh ? 2 : 0, ra_z = t_rh ? 1 : 2, rb_z = t_rh ? 3 : 2;
  const float sgn = t_rh ? -1.f : 1.f;
  const int t_roff = ((2 * t_ty) * HALO_W + t_tx) * W_REC + t_cq * 16;      // (column c of the tile: see tr_read)
  const int t_voff = ((8 * t_rh) * 64 + t_w) * W_REC + ((t_cq ^ ((t_w >> 3) & 1)) * 16);
  auto relu4 = [&](f32x4 f) { return f; };            // (the relu of RELU_IN is applied in LDS: fix_raw)
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
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(zlo), "s"(sgn2), "v"(ylo));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(zhi), "s"(sgn2), "v"(yhi));
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
      if (FISR_WABL & 512) {      // ablation: no LDS reads, opaque register contents instead
        asm volatile("" : "=v"(txa[k]), "=v"(tza[k]), "=v"(tyb[k]), "=v"(tzb[k]));
        continue;
      }
      // halo column 2 tx + c sits at index (c & 1) * 17 + tx + (c >> 1) of its row (even columns first): the 16 lanes
      // of one ds_read_b128 phase -- 8 neighbouring tiles x 2 halves -- then read 256 contiguous bytes, every bank once
      // (with the columns in natural order the tiles are 64 B apart and the phase hits every other bank twice)
      const int cc = (c & 1) * (HALO_W / 2) + (c >> 1);
      txa[k] = *reinterpret_cast<const f32x4*>(rb + (ra_x * HALO_W + cc) * W_REC);
      tza[k] = *reinterpret_cast<const f32x4*>(rb + (ra_z * HALO_W + cc) * W_REC);
      tyb[k] = *reinterpret_cast<const f32x4*>(rb + (1 * HALO_W + cc) * W_REC);
      tzb[k] = *reinterpret_cast<const f32x4*>(rb + (rb_z * HALO_W + cc) * W_REC);
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
  f32x16 acc[8];
  const int f_off = li * W_REC + ((kh ^ ((li >> 3) & 1)) * 16) + (8 * ph) * 64 * W_REC;
  const int fu_off = (32 * nh) * W_REC + f_off;
  const int fv_off = (32 * wh) * W_REC + f_off;
  f32x4 fa[2][2], fb[2][2];
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

  // ---- prologue of the workgroup's FIRST item ----
  int b_cur = blockIdx.x;
  // the copy waves go first where the two waves of a SIMD compete (measured: -2 % on the 256-channel layers, nothing
  // on the 64-channel ones; the other way round: nothing)
  if (ph == 1) __builtin_amdgcn_s_setprio(3);
  Item cur = item_of(b_cur);
  int b_nxt = b_cur + gridDim.x;
  bool has_next = valid(b_nxt);
  Item nxt = has_next ? item_of(b_nxt) : cur;
  if (ph == 1) {
    raw_geom(cur);
    fix_geom(cur);
    u_base = (const char*)p.wpk + (size_t)cur.nblk * W_SLAB;
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

  int par = 0;                                          // V/U buffer of the chunk about to be multiplied
  int slot1 = 1, slot2 = 2, slot3 = 0;                  // RAW slots of (global) chunks g+1, g+2, g+3
  // first k step of an item: C = 0 (an inline constant), so the 128 accumulators are never zeroed by hand
#define FISR_W8_MMA0(PP, RB)                                                                                       \
  acc[2 * (PP)]     = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[RB][0].x, fb[RB][0].x, zero16, 0, 0, 0);             \
  acc[2 * (PP) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[RB][1].x, fb[RB][1].x, zero16, 0, 0, 0);
#define FISR_W8_STAGE0(PP) FISR_W8_MMA0(PP, (PP) & 1) FISR_W8_MMA(PP, (PP) & 1, y) FISR_W8_MMA(PP, (PP) & 1, z) FISR_W8_MMA(PP, (PP) & 1, w)
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  int n_done = 0;
  const int tr_item = max(2, (n_items / (int)gridDim.x) >> 1);     // the item whose timeline is traced (mid-run)
  unsigned long long t_it0 = 0, t_it1 = 0, t_it2 = 0, t_itk = 0, t_bar = 0, t_r0 = 0, t_r2 = 0;

  // The MFMAs of a chunk are skewed by one stage against the barriers: an iteration runs stage 3 of the PREVIOUS chunk
  // (its fragments were loaded into registers before the barrier) and stages 0-2 of its own chunk, and leaves its own
  // stage 3 pending.  A wave issues in order, so without the skew every iteration opens with both waves of a SIMD
  // waiting for their first fragments and the matrix pipe idle for an LDS round trip; with it that round trip hides
  // under eight MFMAs.  The last chunk's stage 3 runs behind the loop (flush_stage3); the first iteration of an item
  // has nothing pending and zeroes the stage-3 accumulators instead (stages 0-2 start from C = 0).
  //
  // slot s of an iteration:  fragments of stage s -> register buffer s & 1;  then the MFMAs of stage (s + 3) & 3 from
  // the other buffer.
#define FISR_W8_SLOT_MMA(S)                                                       \
  if constexpr (FIRST) {                                                          \
    if ((S) == 1) { FISR_W8_STAGE0(0) }                                           \
    if ((S) == 2) { FISR_W8_STAGE0(1) }                                           \
    if ((S) == 3) { FISR_W8_STAGE0(2) }                                           \
  } else {                                                                        \
    if ((S) == 0) { FISR_W8_STAGE(3) }                                            \
    if ((S) == 1) { FISR_W8_STAGE(0) }                                            \
    if ((S) == 2) { FISR_W8_STAGE(1) }                                            \
    if ((S) == 3) { FISR_W8_STAGE(2) }                                            \
  }
  auto zero_stage3 = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[6][r] = 0.f; acc[7][r] = 0.f; }
  };
  auto flush_stage3 = [&]() { FISR_W8_STAGE(3) };

  // one K iteration of a transform wave: chunk kc is multiplied (see above), chunk kc+1 (chunk 0 of the next item behind
  // the last one) is transformed into V[par ^ 1]
  auto iter_transform = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    if (!(FISR_WABL & 1)) tr_read(slot1, 0);
    if constexpr (FIRST) zero_stage3();
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      frag_load(par, sl, sl & 1);
      if (!(FISR_WABL & 1)) {
        if (sl == 0) { tr_rows(0); tr_read(slot1, 1); }
        if (sl == 1) tr_rows(1);
        if (!(FISR_WABL & 1024)) {
          if (sl == 2) tr_cols(par ^ 1, 0);
          if (sl == 3) tr_cols(par ^ 1, 1);
        } else if (sl >= 2) {
          asm volatile("" :: "v"(TA[0]), "v"(TA[1]), "v"(TA[2]), "v"(TA[3]), "v"(TB[0]), "v"(TB[1]), "v"(TB[2]), "v"(TB[3]));
        }
      }
      FISR_W8_SLOT_MMA(sl)
      if (!FIRST || sl > 0) { FISR_W8_INTERLEAVE(4) }
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
    par ^= 1;
  };
  // one K iteration of a copy wave: U(kc+1) and raw(kc+3) are copied; past this item's last chunk the streams continue
  // with the next item's first ones (or, behind the last item, repeat a chunk of this one: the copy COUNT per
  // iteration must stay fixed for the counted waits)
  auto iter_copy = [&](auto first_tag, int kc) {
    constexpr bool FIRST = decltype(first_tag)::value;
    frag_load(par, 0, 0);
    if (kc == nch - 3 && has_next) raw_geom(nxt);                  // raw(kc+3) .. are the next item's from here on
    if (kc == nch - 2 && has_next) fix_geom(nxt);                  // raw(kc+2) is fixed at the end of this iteration
    if (kc == nch - 1 && has_next) u_base = (const char*)p.wpk + (size_t)nxt.nblk * W_SLAB;
    if (!(FISR_WABL & 2)) {
      copy_u(kc + 1 < nch ? kc + 1 : (has_next ? 0 : kc), par ^ 1);
      copy_raw(kc + 3 < nch ? kc + 3 : (has_next ? kc + 3 - nch : nch - 1), slot3);
    }
    if constexpr (FIRST) zero_stage3();
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl > 0) frag_load(par, sl, sl & 1);
      FISR_W8_SLOT_MMA(sl)
      if (sl == 1) {
        // raw(g+2), requested an iteration ago, is older than this iteration's 8 + 3 (2) copies: finish it now, under
        // the MFMAs, not in front of the barrier (read and write halves apart: see relu_read)
        if (!(FISR_WABL & 2)) {
          if (cw < 3) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
          else        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        }
        relu_read(slot2);
      }
      if (sl == 2) { relu_write(slot2); zero_padding(slot2); }
    }
    if (!(FISR_WABL & 2)) wait_copies_keep_youngest_raw();      // U(g+1) landed too; raw(g+3) stays in flight
    lds_barrier();
    const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
    par ^= 1;
  };

  // packed fp32 helpers for the output transform (v_pk_add_f32: two lanes' worth per instruction)
  typedef float f2 __attribute__((ext_vector_type(2)));
#define FISR_W8_PAIR(Q, R) (f2{acc[Q][R], acc[Q][(R) + 1]})
  // column stage of one M row (accumulators QB .. QB+3), output column J, elements R, R+1
  // (asm: the compiler would scalarize the subtraction.  Two textually different copies, one per wave role: identical
  // ones would be merged, hoisted above the role branch and spilled there.)
  auto pk_sub0 = [&](f2 a, f2 b) {
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] ; rows 0, 1" : "=v"(d) : "v"(a), "v"(b));
    return d;
  };
  auto pk_sub1 = [&](f2 a, f2 b) {
    f2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] ; rows 2, 3" : "=v"(d) : "v"(a), "v"(b));
    return d;
  };
#define FISR_W8_COL(SUB, QB, J, R)                                                                                 \
  ((J) == 0 ? (FISR_W8_PAIR(QB, R) + FISR_W8_PAIR((QB) + 1, R)) + FISR_W8_PAIR((QB) + 2, R)                        \
            : SUB(SUB(FISR_W8_PAIR((QB) + 1, R), FISR_W8_PAIR((QB) + 2, R)), FISR_W8_PAIR((QB) + 3, R)))

  for (;;) {                                            // ---- work items of this workgroup ----
    if (p.trace && n_done == tr_item) { t_it0 = __builtin_readcyclecounter(); t_r0 = __builtin_amdgcn_s_memrealtime(); }

    // Epilogue geometry of this item (see conv3x3_wino8.h): lane -> winograd tile (ty, tx), 16 channels from c0.
    // Output and residual are addressed as a per-item 64-bit base (the item's first pixel row) + 32-bit lane offsets.
    // Recomputed from an opaque lane id wherever it is used: hoisted out of the item loop, these per-lane values
    // would stay live across the K loops, which have no register to spare.
    struct Geo { int ty, txq, c0, lq; bool c_ok; };
    auto geometry = [&]() {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int w_ = 32 * wh + (l & 31);
      Geo g;
      g.ty = w_ >> 4;
      g.txq = (w_ & 15) & ~3;
      g.c0 = cur.nblk * W_BN + 32 * nh + 16 * (l >> 5);
      g.c_ok = g.c0 < p.Cout;
      g.lq = l & 3;
      return g;
    };
    // residual records of output row `row` (quad-transposed fetch: lane q of a quad reads bytes 16q.. of each of the
    // quad's four pixels, 64 contiguous bytes per quad and instruction).  Buffer loads: a lane outside the image (or the
    // whole wave, without a residual) gets an offset behind the buffer's end and reads zeros -- no branch per record.
    auto load_res = [&](int row, uint4 (&rres)[2][4]) {      // rres[output column j][pixel k of the quad]
      const Geo g = geometry();
      const unsigned img_bytes = (unsigned)(p.H * p.W) * (unsigned)rec_cs * 4u - (unsigned)rec_co * 4u;
      const unsigned nrec = (p.res != nullptr && !(FISR_WABL & 256)) ? img_bytes : 0u;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)p.res + ((size_t)cur.nb * p.H * p.W * rec_cs + rec_co) * 4), 0, nrec, 0x00020000);
      const int oy = (cur.y0 + 2 * g.ty + row) * dil + cur.ry;
      const bool row_ok = g.c_ok & (oy < p.H);
      const unsigned rowoff = ((unsigned)(oy * p.W) * (unsigned)rec_cs + (unsigned)(g.c0 + 4 * g.lq)) * 4u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = (cur.x0 + 2 * (g.txq + k) + j) * dil + cur.rx;
          const unsigned xo = rowoff + (unsigned)x * (unsigned)rec_cs * 4u;
          const unsigned off = (row_ok & (x < p.W)) ? xo : nrec;     // (&: a select, no control flow)
          rres[j][k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    };

    // ---- K loop ----
    if (ph == 0) {
      iter_transform(first_t{});
      if (p.trace && n_done == tr_item) t_itk = __builtin_readcyclecounter();
      for (int kc = 1; kc < nch; ++kc) iter_transform(rest_t{});
    } else {
      iter_copy(first_t{}, 0);
      if (p.trace && n_done == tr_item) t_itk = __builtin_readcyclecounter();
      for (int kc = 1; kc < nch; ++kc) iter_copy(rest_t{}, kc);
    }
    // residual records of this wave's output row: requested before the last chunk's pending stage, which hides a
    // part of their latency.  (Hoisted further, into the last K iteration, these 32 registers make the compiler spill
    // inside the K loop.)
    uint4 rres[2][4];
    if constexpr (HAS_RES) load_res(ph, rres);
    __builtin_amdgcn_sched_barrier(0);
    flush_stage3();                                // the last chunk's pending stage
    if (p.trace) { t_main = __builtin_readcyclecounter(); if (n_done == tr_item) t_it1 = t_main; }

    // ---- epilogue: Y = A^T M A per winograd tile, + bias, + residual, relu, store.  Wave ph = 0 of a pair holds the M
    //      rows 0 and 1 (accumulators 0-3 and 4-7), wave ph = 1 the rows 2 and 3.  Column stage per row i:
    //          t_i0 = (m_i0 + m_i1) + m_i2          t_i1 = (m_i1 - m_i2) - m_i3
    //      row stage:   output row 0 = (t_0 + t_1) + t_2          output row 1 = (t_1 - t_2) - t_3.
    //      The ph = 1 wave folds the bias and output row 1's residual into what it hands over,
    //          X0 = t_2 + bias           X1 = (bias + res_row1) - (t_2 + t_3)
    //      (through the V / U buffers of the last multiplied chunk, index par ^ 1 after the loop's final toggle: X0 in
    //      V, X1 in U, 8 KB each per wave; V[par] / U[par] already hold the next item's first chunk), and the ph = 0
    //      wave finishes  row 0 = ((t_0 + t_1) + X0) + res_row0,  row 1 = t_1 + X1  and stores both rows.  So the
    //      copy waves (ph = 1) never store: their counted vmcnt waits in the next item would otherwise wait for the
    //      store acknowledgements too (measured: 5k cycles per item).  Tried and slower: all loads and the whole
    //      output stage in the ph = 0 wave (one wave per SIMD then works alone: +3k cycles per item).
    //      In-place residual (res == out): every record is read before the barrier or by the wave that later writes it.
    const int rel = par ^ 1;
    const int xoff = rel * W_SLAB + (wave & 3) * 8192 + lane * 16;
    const Geo geo = geometry();
    const int ty = geo.ty, txq = geo.txq, c0 = geo.c0;
    const bool c_ok = geo.c_ok;
    if (ph == 1) {
      float bv[16];
      {
        const f32x4* bq = reinterpret_cast<const f32x4*>(p.bias + (c_ok ? c0 : 0));
#pragma unroll
        for (int k = 0; k < 4; ++k) { const f32x4 f = bq[k]; bv[4 * k] = f.x; bv[4 * k + 1] = f.y; bv[4 * k + 2] = f.z; bv[4 * k + 3] = f.w; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (HAS_RES) quad_transpose(rres[j], lane);           // -> [k] = this lane's pixel, channels 4k .. 4k+3
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x4 r1 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (HAS_RES) r1 = __builtin_bit_cast(f32x4, rres[j][k]);
          f32x4 X0, X1;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int r = 4 * k + e;
            const f2 t2 = FISR_W8_COL(pk_sub1, 0, j, r), t3 = FISR_W8_COL(pk_sub1, 4, j, r);
            const f2 b = {bv[r], bv[r + 1]};
            const f2 x0 = t2 + b;
            const f2 x1 = pk_sub1(HAS_RES ? b + f2{r1[e], r1[e + 1]} : b, t2 + t3);
            X0[e] = x0.x; X0[e + 1] = x0.y; X1[e] = x1.x; X1[e + 1] = x1.y;
          }
          *reinterpret_cast<f32x4*>(sV + xoff + (j * 4 + k) * 1024) = X0;
          *reinterpret_cast<f32x4*>(sU + xoff + (j * 4 + k) * 1024) = X1;
        }
      }
    }
    lds_barrier();
    if (p.trace && n_done == tr_item) t_bar = __builtin_readcyclecounter();
    if (ph == 0) {
      // Store addressing without a branch per record: inside image nb the byte offset of the lane's 16 bytes is
      //     4 * (oy * A + x * B + C),   A, B uniform, C per lane
      // (depth_to_space: record (y, x), channel c0 -> pixel (2y + sub / 2, 2x + sub % 2) of the 2H x 2W image, channel
      // c0 % cq with sub = c0 / cq, cq = Cout / 4), and buffer stores, which drop the lanes whose offset lies behind the
      // buffer's end: that is where the lanes outside the image point.  The relu is a maximum with 0 or -inf (leaky:
      // with slope * v or 1 * v), a uniform operand instead of a branch.  (With a branch per record and flag the 16
      // stores of a wave were 100 taken or skipped branches: 2k of the 6k cycles of this stage, which one wave per SIMD
      // runs alone.)
      const int cq_shift = p.d2s_shift;
      const unsigned sub = (unsigned)c0 >> cq_shift;
      const unsigned sA = p.d2s ? (unsigned)(4 * p.W) << cq_shift : (unsigned)p.W * (unsigned)rec_cs;
      const unsigned sB = p.d2s ? 2u << cq_shift : (unsigned)rec_cs;
      const unsigned vC = (p.d2s ? ((((sub >> 1) * 2u * (unsigned)p.W + (sub & 1u)) << cq_shift) + ((unsigned)c0 & ((1u << cq_shift) - 1u)))
                                 : (unsigned)c0) + 4u * (unsigned)geo.lq;
      const unsigned out_bytes = p.d2s ? ((unsigned)(4 * p.H * p.W) << cq_shift) * 4u
                                       : (unsigned)(p.H * p.W) * (unsigned)rec_cs * 4u - (unsigned)rec_co * 4u;
      const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
          (char*)p.out + (p.d2s ? ((size_t)cur.nb * 2 * p.H * 2 * p.W << cq_shift) * 4
                                : ((size_t)cur.nb * p.H * p.W * rec_cs + rec_co) * 4), 0, out_bytes, 0x00020000);
      const float relu_lo = __builtin_bit_cast(float, p.relu_out ? 0u : 0xff800000u);
      const float relu_sl = p.relu_out ? slope : 1.f;
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int j = 0; j < 2; ++j) {              // output column j of every winograd tile
        uint4 rec[2][4];
        if constexpr (HAS_RES) quad_transpose(rres[j], lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) {            // channels 4k .. 4k+3 of the record
          const f32x4 X0 = *reinterpret_cast<const f32x4*>(sV + xoff + (j * 4 + k) * 1024);
          const f32x4 X1 = *reinterpret_cast<const f32x4*>(sU + xoff + (j * 4 + k) * 1024);
          f32x4 r0 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (HAS_RES) r0 = __builtin_bit_cast(f32x4, rres[j][k]);
          f32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int r = 4 * k + e;
            const f2 t0 = FISR_W8_COL(pk_sub0, 0, j, r), t1 = FISR_W8_COL(pk_sub0, 4, j, r);
            f2 y0 = (t0 + t1) + f2{X0[e], X0[e + 1]};
            if constexpr (HAS_RES) y0 += f2{r0[e], r0[e + 1]};
            const f2 y1 = t1 + f2{X1[e], X1[e + 1]};
            o0[e] = y0.x; o0[e + 1] = y0.y; o1[e] = y1.x; o1[e + 1] = y1.y;
          }
          if (GENERAL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = fmaxf(o0[e], relu_sl * o0[e]); o1[e] = fmaxf(o1[e], relu_sl * o1[e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              asm("v_max_f32 %0, %1, %0" : "+v"(o0[e]) : "s"(relu_lo));
              asm("v_max_f32 %0, %1, %0" : "+v"(o1[e]) : "s"(relu_lo));
            }
          }
          rec[0][k] = __builtin_bit_cast(uint4, o0);
          rec[1][k] = __builtin_bit_cast(uint4, o1);
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          quad_transpose(rec[row], lane);
          const int oy = (cur.y0 + 2 * ty + row) * dil + cur.ry;
          const bool row_ok = c_ok & (oy < p.H);
          const unsigned rowoff = ((unsigned)oy * sA + vC) * 4u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = (cur.x0 + 2 * (txq + k) + j) * dil + cur.rx;
            const unsigned xo = rowoff + (unsigned)x * sB * 4u;
            const unsigned off = (row_ok & (x < p.W) & !(FISR_WABL & 128)) ? xo : out_bytes;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rec[row][k]), os, off, 0, 2);   // aux 2: nontemporal
          }
        }
      }
    }
    if (p.trace && n_done == tr_item) { t_it2 = __builtin_readcyclecounter(); t_r2 = __builtin_amdgcn_s_memrealtime(); }
    ++n_done;
    if (!has_next) break;
    lds_barrier();                               // the handed-over rows are read before the next item overwrites V / U
    b_cur = b_nxt; cur = nxt;
    b_nxt = b_cur + gridDim.x;
    has_next = valid(b_nxt);
    nxt = has_next ? item_of(b_nxt) : cur;
  }
#undef FISR_W8_PAIR
#undef FISR_W8_COL
#undef FISR_W8_MMA0
#undef FISR_W8_SLOT_MMA
#undef FISR_W8_STAGE0
#undef FISR_W8_MMA
#undef FISR_W8_STAGE
#undef FISR_W8_INTERLEAVE
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    // with three or more items: the timeline of the THIRD one (steady state; "prologue" = 0); else the whole life
    const bool third = n_done > tr_item;
    tr[0] = third ? t_it0 : t_start; tr[1] = third ? t_it1 : t_main; tr[2] = third ? t_it2 : __builtin_readcyclecounter();
    tr[3] = t_bar;
    tr[4] = third ? t_itk : t_first; tr[5] = third ? t_r0 : t_real; tr[6] = third ? t_r2 : __builtin_amdgcn_s_memrealtime();
    tr[7] = (unsigned long long)n_done;
  }
}

}  // namespace fisr
