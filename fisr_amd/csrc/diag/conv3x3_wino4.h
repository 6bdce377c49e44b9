// The first Winograd F(2x2,3x3) kernel of this build: 256 threads, one wave per SIMD, all 16 transform positions of a
// wave's 32 wtiles x 32 channels in its own registers (256 accumulator registers, no exchange in the output transform).
// Superseded by conv3x3_wino8p.h; kept for A/B runs, compiled into -DFISR_DIAG builds only (scripts/gpu_wino.sh).
#pragma once
#include "../conv3x3_wino_common.h"

namespace fisr {

__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sV = smem;
  char* const sU = smem + 2 * W_SLAB;
  char* const sR = smem + 4 * W_SLAB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int kh = lane >> 5;
  const int wh = wave & 1;     // which 32 wtiles (pixel rows 0-3 / 4-7 of the tile)
  const int nh = wave >> 1;    // which 32 of the 64 output channels

  const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
  const int tiles_y = (p.H + TILE_H - 1) / TILE_H;
  // XCD-aware work order, as in conv3x3.h (speed only): contiguous virtual ids per XCD, the N-blocks of one
  // pixel tile consecutive on it.
  int v = blockIdx.x;
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

  // ---- raw halo chunk: 680 16-byte units, unit u = tid + 256*i -> halo pixel u >> 1, half u & 1 (= tid & 1) ----
  // Every unit is copied from a CLAMPED address (padding and beyond-the-image units read some valid pixel), so the
  // number of copy instructions per wave is fixed -- the counted waits rely on it -- and the units that must be
  // zero (SAME padding, ragged edges) are overwritten with zeros by the same thread once the copy has landed.
  // Waves 0, 1: three full copies; wave 2: two full + lanes 0..39; wave 3: two.
  unsigned raw_voff0[3], raw_voff1[3];                  // byte offsets into in0 / in1 (32-bit: tensors < 4 GB, host-checked)
  bool in_ok[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = tid + 256 * i;
    const int pix = min(u >> 1, HALO_PIX - 1);
    const int py = pix / HALO_W, px = pix - py * HALO_W;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    in_ok[i] = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;       // (units >= 680 are never copied nor fixed)
    const unsigned gp = (unsigned)((nb * p.H + min(max(gy, 0), p.H - 1)) * p.W + min(max(gx, 0), p.W - 1));
    raw_voff0[i] = (gp * (unsigned)p.C0 + (unsigned)(tid & 1) * 4u) * 4u;
    raw_voff1[i] = (gp * (unsigned)p.C1 + (unsigned)(tid & 1) * 4u) * 4u;
  }
  const unsigned raw_lds0 = (unsigned)(size_t)(lds_ptr_t)sR + (unsigned)wave * 1024u;
  auto copy_raw = [&](int kc, int slot) {
    const int c0 = min(kc, nch - 1) * W_CH;          // past the last chunk: a harmless repeat of the last one
    const bool first = c0 < p.C0;
    const char* g = first ? (const char*)p.in0 + (size_t)c0 * 4 : (const char*)p.in1 + (size_t)(c0 - p.C0) * 4;
    const unsigned o0 = first ? raw_voff0[0] : raw_voff1[0];
    const unsigned o1 = first ? raw_voff0[1] : raw_voff1[1];
    const unsigned o2 = first ? raw_voff0[2] : raw_voff1[2];
    const unsigned lds = raw_lds0 + (unsigned)slot * (unsigned)W_RAW;
    unsigned keep;
    if (wave < 2) {
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g)
                   FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o2, g) FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2) : "memory", "scc");
    } else if (wave == 2) {
      unsigned long long ex;
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g)
                   FISR_GLDS_NEXT_ROW
                   "s_mov_b64 %[ex], exec\n\ts_bfm_b64 exec, 40, 0\n\t"      // units 640..679: lanes 0..39 of wave 2
                   FISR_GLDS_COPY(o2, g)
                   "s_mov_b64 exec, %[ex]\n\t" FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep), [ex] "=&s"(ex) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2)
                   : "memory", "scc");
    } else {
      asm volatile(FISR_GLDS_BEGIN(keep, lds) FISR_GLDS_COPY(o0, g) FISR_GLDS_NEXT_ROW FISR_GLDS_COPY(o1, g) FISR_GLDS_END(keep)
                   : [keep] "=&s"(keep) : [g] "s"(g), [lds] "s"(lds), [o0] "v"(o0), [o1] "v"(o1) : "memory", "scc");
    }
  };
  // zeros over the units of a landed raw chunk that are padding (by the thread that copied them)
  const bool any_pad = !(in_ok[0] && in_ok[1] && (in_ok[2] || tid + 512 >= W_RAW_UNITS));
  auto fix_raw = [&](int slot) {
    if (any_pad) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int u = tid + 256 * i;
        if (!in_ok[i] && u < W_RAW_UNITS) *reinterpret_cast<f32x4*>(sR + slot * W_RAW + u * 16) = z;
      }
    }
  };

  // ---- U slab: 2048 16-byte units, a linear copy of the host-made LDS image of (chunk kc, N-block) ----
  const char* const u_base = (const char*)p.wpk + (size_t)nblk * W_SLAB;
  const size_t u_stride = (size_t)nblocks * W_SLAB;
  const unsigned u_lds0 = (unsigned)(size_t)(lds_ptr_t)sU + (unsigned)wave * 1024u;   // LDS byte address, this wave's 1 KB
  const unsigned u_voff = (unsigned)tid * 16u;
  auto copy_u = [&](int kc, int buf) {
    const char* g = u_base + (size_t)kc * u_stride;                 // wave-uniform: SGPR pair
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)W_SLAB;
    unsigned keep;
    // unit (tid + 256*i), i = 0..7: global byte offset voff + 4096*i, LDS destination M0 + lane*16
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
  // Waits and the workgroup barrier.  __syncthreads() is a release fence (vmcnt(0)); here a wave waits for all its
  // copies EXCEPT the youngest raw chunk's (copies complete in order; the raw copies of the chunk three ahead are
  // issued after the U copy precisely so that they may stay in flight across the barrier), fixes the padding of the
  // chunk that has just landed, waits for its own LDS writes and joins the barrier.
  auto wait_copies_keep_youngest_raw = [&]() {
    if (wave < 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- input transform V = B^T d B: thread = (wtile t_w, channel quad t_cq, row half t_rh) ----
  // B^T rows: T0 = d0 - d2, T1 = d1 + d2, T2 = d2 - d1, T3 = d1 - d3 (then the same along columns).  Row half 0
  // makes T0, T1 of its wtile, row half 1 (waves 2, 3) T2, T3; per wave the row roles are uniform:
  //   A = d[ra_x] - d[ra_z]        B = d[1] + sgn * d[rb_z]
  const int t_cq = tid & 1, t_w = (tid >> 1) & 63;
  const int t_rh = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int t_ty = t_w >> 4, t_tx = t_w & 15;
  const int ra_x = t_rh ? 2 : 0, ra_z = t_rh ? 1 : 2, rb_z = t_rh ? 3 : 2;
  const float sgn = t_rh ? -1.f : 1.f;
  const int t_roff = ((2 * t_ty) * HALO_W + 2 * t_tx) * W_REC + t_cq * 16;
  const int t_voff = ((8 * t_rh) * 64 + t_w) * W_REC + ((t_cq ^ ((t_w >> 3) & 1)) * 16);
  const float relu_in_floor = p.relu_in ? 0.f : -__builtin_huge_valf();   // branch-free relu-on-load
  auto relu4 = [&](f32x4 f) {
    f.x = fmaxf(f.x, relu_in_floor); f.y = fmaxf(f.y, relu_in_floor);
    f.z = fmaxf(f.z, relu_in_floor); f.w = fmaxf(f.w, relu_in_floor);
    return f;
  };
  // The transform is cut in five slices so that it can be spread between the MFMAs of the running chunk:
  //   slice 0: raw reads of columns 0,1      slice 1: raw reads of columns 2,3 + row stage of columns 0,1
  //   slice 2: row stage of columns 2,3      slice 3: column stage + stores of T-row A
  //   slice 4: column stage + stores of T-row B
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
      TA[2 * cpair + k] = relu4(txa[k]) - relu4(tza[k]);
      TB[2 * cpair + k] = relu4(tyb[k]) + sgn * relu4(tzb[k]);
    }
  };
  auto tr_cols = [&](int vbuf, int which) {
    char* vw = sV + vbuf * W_SLAB + t_voff + which * 4 * 64 * W_REC;
    const f32x4* T = which ? TB : TA;
    *reinterpret_cast<f32x4*>(vw + 0 * 64 * W_REC) = T[0] - T[2];
    *reinterpret_cast<f32x4*>(vw + 1 * 64 * W_REC) = T[1] + T[2];
    *reinterpret_cast<f32x4*>(vw + 2 * 64 * W_REC) = T[2] - T[1];
    *reinterpret_cast<f32x4*>(vw + 3 * 64 * W_REC) = T[1] - T[3];
  };

  // ---- MFMA phase: for every position, 8 channels = one 16-byte fragment per operand = 4 MFMAs (K = 2) ----
  // lane (li, kh): A operand (rows) = U record of channel row 32*nh + li, B operand (cols) = V record of wtile
  // 32*wh + li; 16-byte half kh (channels 4*kh .. 4*kh+3), stored swizzled by bit 3 of the record index.
  // A chunk is 8 stages of 2 positions (8 MFMAs = 512 cycles of the SIMD's matrix pipe); the fragments of
  // stage s+1 are requested before the MFMAs of stage s (one wave per SIMD: nobody else hides LDS latency).
  f32x16 acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int f_off = li * W_REC + ((kh ^ ((li >> 3) & 1)) * 16);
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
#define FISR_WINO_MMA(PP, RB, E)                                                                                   \
  acc[2 * (PP)]     = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[RB][0].E, fb[RB][0].E, acc[2 * (PP)], 0, 0, 0);      \
  acc[2 * (PP) + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[RB][1].E, fb[RB][1].E, acc[2 * (PP) + 1], 0, 0, 0);
  // interleave request for the scheduler: every MFMA is followed by up to NOTHER non-matrix instructions
  // (VALU / LDS reads and writes of the transform slice and of the next fragments)
#define FISR_WINO_INTERLEAVE(NOTHER)                                                  \
  _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
    __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200, NOTHER, 0);           \
  }

  // ---- prologue: U(0) and raw(0..2) requested; chunk 0 transformed; raw(1) fixed ----
  // raw chunk c lives in RAW[c % 3]: requested at iteration c-3, landed + padding fixed at the end of iteration
  // c-2, transformed during iteration c-1 (into V[c & 1]), multiplied in iteration c.
  copy_raw(0, 0);
  copy_u(0, 0);
  copy_raw(1, 1);
  copy_raw(2, 2);
  if (wave < 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // raw(0), U(0) landed; raw(1), raw(2) in flight
  else          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  fix_raw(0);
  lds_barrier();
  tr_read(0, 0); tr_rows(0); tr_read(0, 1); tr_rows(1); tr_cols(0, 0); tr_cols(0, 1);     // chunk 0 -> V[0]
  wait_copies_keep_youngest_raw();                                  // raw(1) landed; raw(2) in flight
  fix_raw(1);
  lds_barrier();
  if (p.trace) t_first = __builtin_readcyclecounter();

  // ---- main loop: MFMAs of chunk kc || input transform of chunk kc+1 || copies of U(kc+1), raw(kc+3) ----
  int slot1 = 1, slot2 = 2, slot3 = 0;                 // RAW slots of chunks kc+1, kc+2, kc+3
  for (int kc = 0; kc + 1 < nch; ++kc) {
    const int b = kc & 1;
    if (!(FISR_WABL & 2)) {
      copy_u(kc + 1, b ^ 1);
      copy_raw(kc + 3, slot3);
    }
    frag_load(b, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      if (pp < 7 && !(FISR_WABL & 8)) frag_load(b, pp + 1, (pp + 1) & 1);
      if (!(FISR_WABL & 1)) {
        if (pp == 0) tr_read(slot1, 0);
        if (pp == 1) { tr_rows(0); tr_read(slot1, 1); }
        if (pp == 2) tr_rows(1);
        if (pp == 3) tr_cols(b ^ 1, 0);
        if (pp == 4) tr_cols(b ^ 1, 1);
      }
      if (!(FISR_WABL & 4)) {
        FISR_WINO_MMA(pp, (FISR_WABL & 8) ? 0 : (pp & 1), x) FISR_WINO_MMA(pp, (FISR_WABL & 8) ? 0 : (pp & 1), y)
        FISR_WINO_MMA(pp, (FISR_WABL & 8) ? 0 : (pp & 1), z) FISR_WINO_MMA(pp, (FISR_WABL & 8) ? 0 : (pp & 1), w)
      }
      if (!(FISR_WABL & 16)) { FISR_WINO_INTERLEAVE(3) }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(FISR_WABL & 2)) wait_copies_keep_youngest_raw();      // U(kc+1) and raw(kc+2) landed; raw(kc+3) stays in flight
    if (!(FISR_WABL & 32)) fix_raw(slot2);
    lds_barrier();
    const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
  }

  // epilogue geometry: lane (li, kh) owns wtile 32*wh + li -> pixels (y0 + 2*ty + i, x0 + 2*tx + j), and the
  // 16-channel record c0 .. c0+15.  The residual records (4 pixels x 64 B) and the bias are requested BEFORE the
  // last chunk's MFMAs so their latency hides under them (the residual may alias the output: every element is
  // read and written by the same lane only, so hoisting the reads above the stores is safe).  These are the only
  // compiler-tracked global loads of the kernel; the hidden copies still in flight (a repeat of the last raw
  // chunk) are OLDER, so the compiler's counted waits for them stay conservative.
  const int w_ = 32 * wh + li;
  const int ty = w_ >> 4, tx = w_ & 15;
  const int c0 = n0 + 32 * nh + 16 * kh;
  const bool c_ok = c0 < p.Cout;
  uint4 rres[2][2][4];
  float bv[16];
  bool px_ok[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = y0 + 2 * ty + i, x = x0 + 2 * tx + j;
      px_ok[i][j] = c_ok && y < p.H && x < p.W;
#pragma unroll
      for (int k = 0; k < 4; ++k) rres[i][j][k] = make_uint4(0u, 0u, 0u, 0u);
      if (p.res != nullptr && px_ok[i][j]) {
        const uint4* q = reinterpret_cast<const uint4*>((const float*)p.res + ((size_t)(nb * p.H + y) * p.W + x) * p.Cout + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k) rres[i][j][k] = q[k];
      }
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
    for (int pp = 0; pp < 8; ++pp) {
      if (pp < 7) frag_load(b, pp + 1, (pp + 1) & 1);
      FISR_WINO_MMA(pp, pp & 1, x) FISR_WINO_MMA(pp, pp & 1, y) FISR_WINO_MMA(pp, pp & 1, z) FISR_WINO_MMA(pp, pp & 1, w)
    }
  }
#undef FISR_WINO_MMA
#undef FISR_WINO_INTERLEAVE
  if (p.trace) t_main = __builtin_readcyclecounter();

  // ---- epilogue: output transform A^T M A per lane, + bias (+ residual), relu, 64-byte records ----
  // One output-tile row i at a time and four channels at a time: few accumulators are in flight between the
  // accumulator file and the VALU, so nothing spills.
  {
    const float relu_floor = p.relu_out ? 0.f : -__builtin_huge_valf();
    const int cq_shift = p.d2s_shift;
    auto record = [&](int y, int xc) -> size_t {     // first output element of this lane's record at pixel (y, xc)
      if (p.d2s) {
        const int sub = c0 >> cq_shift, c = c0 & ((1 << cq_shift) - 1);
        return (((size_t)(nb * 2 * p.H + 2 * y + (sub >> 1))) * (2 * p.W) + 2 * xc + (sub & 1)) * ((size_t)1 << cq_shift) + c;
      }
      return ((size_t)(nb * p.H + y) * p.W + xc) * p.Cout + c0;
    };
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4_t* dst0 = reinterpret_cast<u32x4_t*>((float*)p.out + (px_ok[i][0] ? record(y0 + 2 * ty + i, x0 + 2 * tx) : 0));
      u32x4_t* dst1 = reinterpret_cast<u32x4_t*>((float*)p.out + (px_ok[i][1] ? record(y0 + 2 * ty + i, x0 + 2 * tx + 1) : 0));
#pragma unroll
      for (int k = 0; k < 4; ++k) {                  // channels 4k .. 4k+3 of the record
        f32x4 o0, o1;
        const f32x4 r0 = __builtin_bit_cast(f32x4, rres[i][0][k]), r1 = __builtin_bit_cast(f32x4, rres[i][1][k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * k + e;
          // s_c = column c of row i of (A^T M):  i = 0: m0c + m1c + m2c,  i = 1: m1c - m2c - m3c
          float sc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            sc[c] = i == 0 ? (acc[0 + c][r] + acc[4 + c][r]) + acc[8 + c][r]
                           : (acc[4 + c][r] - acc[8 + c][r]) - acc[12 + c][r];
          const float y0v = (FISR_WABL & 64) ? acc[e][r] : (sc[0] + sc[1]) + sc[2];
          const float y1v = (FISR_WABL & 64) ? acc[4 + e][r] : (sc[1] - sc[2]) - sc[3];
          o0[e] = fmaxf((y0v + bv[r]) + r0[e], relu_floor);
          o1[e] = fmaxf((y1v + bv[r]) + r1[e], relu_floor);
        }
        if (px_ok[i][0]) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, o0), dst0 + k);
        if (px_ok[i][1]) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, o1), dst1 + k);
        __builtin_amdgcn_sched_barrier(0);
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
