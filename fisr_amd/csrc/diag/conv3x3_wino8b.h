// Winograd F(2x2,3x3) convolution of fp32 tensors with the PRODUCTS on the 16-bit matrix pipe: the transformed operands are
// split into bf16 pairs, U = Uh + Ul (host, from U = G g G^T in double), V = Vh + Vl (in the kernel, behind the fp32 input
// transform), and every product is (Uh + Ul)(Vh + Vl) accumulated in fp32 -- 16 of the direct algorithm's 36 multiplies, each
// costing 4/16 of an fp32 MFMA's time (r06; reference operator: ops.py:7-11 + the fused neighbours of conv3x3.h).
//
// Same tensors (NHWC fp32), same work items, same LDS plan, same copy / relu-fix / epilogue code as conv3x3_wino8p.h -- that
// file's comments describe them; what differs is the K loop's arithmetic:
//   * V and U records stay 32 bytes per (position, tile | output channel) and 8-channel chunk, now {8 x bf16 hi | 8 x bf16 lo}
//     (the two 16-byte halves swapped when bit 3 of the record index is set, as before);
//   * one K = 16 MFMA covers the chunk's 8 channels twice: A = [Uh | Ul] (lane half kh reads the record's half kh: one
//     ds_read_b128), B = [Vh | Vh] and then [Vl | Vl] (both lane halves read the same half):
//         acc += Uh Vh + Ul Vh;   acc += Uh Vl + Ul Vl
//     two v_mfma_f32_32x32x16_bf16 (64 cycles) per position and 32 x 32 block where the fp32 kernel spends four
//     v_mfma_f32_32x32x2_f32 (256 cycles);
//   * the transform waves round each V value to bf16 (RNE), subtract, round the remainder: |V - Vh - Vl| <= 2^-17 |V|.
// A wave holds 8 positions x 32 channels x 32 tiles as in the fp32 kernel, so the output stage is that kernel's, unchanged.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>
#include "../conv3x3_wino8p.h"      // (the shared constants and LDS-DMA macros; first_t / rest_t)

namespace fisr {

// -DFISR_WB_TRACE (scripts/probes/winob_bench.hip only): per wave, the cycles of an iteration's phases summed over one launch --
// p.trace[(workgroup * 8 + wave) * 8 + {0: first phase (MFMA | copies + fix), 1: second phase (transform | MFMA), 2: barrier wait,
// 3: iterations, 4: epilogue, 5: whole life}]
#ifdef FISR_WB_TRACE
#define FISR_WB_T0 unsigned long long tq_ = __builtin_readcyclecounter();
#define FISR_WB_T1(I) { const unsigned long long tn_ = __builtin_readcyclecounter(); tph[I] += tn_ - tq_; tq_ = tn_; }
#else
#define FISR_WB_T0
#define FISR_WB_T1(I)
#endif

template <bool RELU_IN, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void conv3x3_wino8b_kernel(const ConvArgs p, const int n_items) {
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
  const int nblocks = p.CoutPad / W_BN;
  const int nch = (p.C0 + p.C1) / W_CH;
#ifdef FISR_WB_TRACE
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long t_life0 = __builtin_readcyclecounter();
#endif

  // work item b -> (x0, y0, nb, nblk): the XCD-aware order of conv3x3_wino8p.h
  struct Item { int x0, y0, nb, nblk; };
  auto item_of = [&](int b) {
    const int q = n_items >> 3, r = n_items & 7;
    const int xcd = b & 7, loc = b >> 3;
    int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int t = v / nblocks;
    Item it;
    it.nblk = v - t * nblocks;
    const int tx_ = t % tiles_x; t /= tiles_x;
    const int ty_ = t % tiles_y; t /= tiles_y;
    it.nb = t;
    it.x0 = tx_ * TILE_W; it.y0 = ty_ * TILE_H;
    return it;
  };
  auto valid = [&](int b) { return b < n_items; };

  // =========================== copy side (waves 4-7) -- conv3x3_wino8p.h ===========================
  const int ct = tid & 255;
  const int cw = wave & 3;
  unsigned raw_gp[3];
  bool fix_ok[3];
  bool fix_any = false;
  const char* u_base = nullptr;
  auto raw_geom = [&](const Item& it) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int u = ct + 256 * i;
      asm volatile("" : "+v"(u));
      const int pix = min(u >> 1, HALO_PIX - 1);
      const int py = pix / HALO_W, ix = pix - py * HALO_W;
      const int px = ix < HALO_W / 2 ? 2 * ix : 2 * (ix - HALO_W / 2) + 1;     // even columns first, then the odd ones
      const int gy = it.y0 - 1 + py, gx = it.x0 - 1 + px;
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
      const int px = ix < HALO_W / 2 ? 2 * ix : 2 * (ix - HALO_W / 2) + 1;
      const int gy = it.y0 - 1 + py, gx = it.x0 - 1 + px;
      fix_ok[i] = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) || u >= W_RAW_UNITS;
      fix_any = fix_any || !fix_ok[i];
    }
  };
  const unsigned raw_lds0 = (unsigned)(size_t)(lds_ptr_t)sR + (unsigned)cw * 1024u;
  auto copy_raw = [&](int kc, int slot) {
    const int c0 = kc * W_CH;
    const bool first = c0 < p.C0;
    const char* g = first ? (const char*)p.in0 + (size_t)c0 * 4 : (const char*)p.in1 + (size_t)(c0 - p.C0) * 4;
    const unsigned cs = (unsigned)(first ? p.C0 : p.C1) * 4u, ho = (unsigned)(ct & 1) * 16u;
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
  auto copy_u = [&](int kc, int buf) {
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
  auto wait_copies_keep_youngest_raw = [&]() {
    if (cw < 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // =========================== transform side (waves 0-3) ===========================
  // thread -> (channel quad t_cq of the chunk, tile t_w, row half t_rh) as in conv3x3_wino8p.h; the eight V values of a thread
  // (two T rows x four columns, four channels each) are split into bf16 hi / lo and stored as two 8-byte pieces per value quad
  const int t_cq = tid & 1, t_w = (tid >> 1) & 63;
  const int t_rh = __builtin_amdgcn_readfirstlane((tid >> 7) & 1);
  const int t_ty = t_w >> 4, t_tx = t_w & 15;
  const int ra_x = t_rh ? 2 : 0, ra_z = t_rh ? 1 : 2, rb_z = t_rh ? 3 : 2;
  const float sgn = t_rh ? -1.f : 1.f;
  const int t_roff = ((2 * t_ty) * HALO_W + t_tx) * W_REC + t_cq * 16;
  const int t_voff = ((8 * t_rh) * 64 + t_w) * W_REC + (((t_w >> 3) & 1) * 16) + t_cq * 8;      // the hi piece; lo: ^ 16
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
      TA[2 * cpair + k] = sub4(txa[k], tza[k]);
      TB[2 * cpair + k] = fma4_sgn(tzb[k], tyb[k]);
    }
  };
  // (a second register set for the other column pair: all 16 raw reads of a chunk are in flight before the first one is used)
  f32x4 txa2[2], tza2[2], tyb2[2], tzb2[2];
  auto tr_read2 = [&](int slot, int cpair) {
    const char* rb = sR + slot * W_RAW + t_roff;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = 2 * cpair + k;
      const int cc = (c & 1) * (HALO_W / 2) + (c >> 1);
      txa2[k] = *reinterpret_cast<const f32x4*>(rb + (ra_x * HALO_W + cc) * W_REC);
      tza2[k] = *reinterpret_cast<const f32x4*>(rb + (ra_z * HALO_W + cc) * W_REC);
      tyb2[k] = *reinterpret_cast<const f32x4*>(rb + (1 * HALO_W + cc) * W_REC);
      tzb2[k] = *reinterpret_cast<const f32x4*>(rb + (rb_z * HALO_W + cc) * W_REC);
    }
  };
  auto tr_rows2 = [&](int cpair) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      TA[2 * cpair + k] = sub4(txa2[k], tza2[k]);
      TB[2 * cpair + k] = fma4_sgn(tzb2[k], tyb2[k]);
    }
  };
  // v -> {bf16(v), bf16(v - bf16(v))}, both round-to-nearest-even (v_cvt_pk_bf16_f32), as two 8-byte LDS pieces
  auto split_store = [&](int voff, f32x4 v) {      // voff: the hi piece's offset inside sV; the lo piece sits at voff ^ 16
    const uint32_t h01 = cvt_pk_bf16(v.x, v.y), h23 = cvt_pk_bf16(v.z, v.w);
    const f32x4 hf = {__builtin_bit_cast(float, h01 << 16), __builtin_bit_cast(float, h01 & 0xffff0000u),
                      __builtin_bit_cast(float, h23 << 16), __builtin_bit_cast(float, h23 & 0xffff0000u)};
    const f32x4 r = sub4(v, hf);
    const uint32_t l01 = cvt_pk_bf16(r.x, r.y), l23 = cvt_pk_bf16(r.z, r.w);
    *reinterpret_cast<uint2*>(sV + voff) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(sV + (voff ^ 16)) = make_uint2(l01, l23);
  };
  // column stage of T row `which` (TA | TB): its positions j = 2 half, 2 half + 1
  auto tr_cols = [&](int vbuf, int which, int half) {
    const int vw = vbuf * W_SLAB + t_voff + which * 4 * 64 * W_REC;
    const f32x4* T = which ? TB : TA;
    if (half == 0) {
      split_store(vw + 0 * 64 * W_REC, sub4(T[0], T[2]));
      split_store(vw + 1 * 64 * W_REC, add4(T[1], T[2]));
    } else {
      split_store(vw + 2 * 64 * W_REC, sub4(T[2], T[1]));
      split_store(vw + 3 * 64 * W_REC, sub4(T[1], T[3]));
    }
  };

  // =========================== MFMA side (all waves) ===========================
  f32x16 acc[8];
  const int swz = ((li >> 3) & 1) * 16;
  const int fu_off = ((8 * ph) * 64 + 32 * nh + li) * W_REC + ((kh * 16) ^ swz);
  const int fvh_off = ((8 * ph) * 64 + 32 * wh + li) * W_REC + swz;
  const int fvl_off = fvh_off ^ 16;
  constexpr int RD = 4;                // positions of fragments in flight (12 registers each)
  bf16x8 fa[RD], fvh[RD], fvl[RD];
  auto frag_load = [&](int buf, int s, int r) {
    fa[r] = *reinterpret_cast<const bf16x8*>(sU + buf * W_SLAB + fu_off + s * 64 * W_REC);
    fvh[r] = *reinterpret_cast<const bf16x8*>(sV + buf * W_SLAB + fvh_off + s * 64 * W_REC);
    fvl[r] = *reinterpret_cast<const bf16x8*>(sV + buf * W_SLAB + fvl_off + s * 64 * W_REC);
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // position s of the wave: acc[s] (+)= [Uh | Ul] [Vh | Vh] + [Uh | Ul] [Vl | Vl]; the first chunk of an item starts from C = 0
  // (two positions at a time, so that a dependent MFMA never directly follows its producer: back to back on one accumulator the
  //  pair costs 48 instead of 32 cycles each with two waves on the SIMD -- scripts/probes/mfma_bf16_k8_rate_probe.hip)
#define FISR_W8B_MMA2(S, R0, R1)                                                                                         \
  acc[S] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[R0], fvh[R0], FIRST ? zero16 : acc[S], 0, 0, 0);                   \
  acc[(S) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[R1], fvh[R1], FIRST ? zero16 : acc[(S) + 1], 0, 0, 0);       \
  acc[S] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[R0], fvl[R0], acc[S], 0, 0, 0);                                    \
  acc[(S) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[R1], fvl[R1], acc[(S) + 1], 0, 0, 0);
  // the MFMA phase of an iteration: the chunk's 8 positions, fragments RD positions ahead
#define FISR_W8B_MFMA_PHASE()                                                                                            \
  _Pragma("unroll") for (int s_ = 0; s_ < RD; ++s_) frag_load(par, s_, s_);                                              \
  _Pragma("unroll") for (int s_ = 0; s_ < 8; s_ += 2) {                                                                  \
    FISR_W8B_MMA2(s_, s_ % RD, (s_ + 1) % RD)                                                                            \
    if (s_ + RD < 8) { frag_load(par, s_ + RD, s_ % RD); frag_load(par, s_ + RD + 1, (s_ + 1) % RD); }                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }

  // ---- prologue of the workgroup's FIRST item ----
  int b_cur = blockIdx.x;
  // -DFISR_WB_PRIO: 1 (default, as conv3x3_wino8p.h) the copy waves go first where the two waves of a SIMD compete, 2 the
  // transform waves, 0 nobody
#ifndef FISR_WB_PRIO
#define FISR_WB_PRIO 1
#endif
  if (FISR_WB_PRIO == 1 && ph == 1) __builtin_amdgcn_s_setprio(3);
  if (FISR_WB_PRIO == 2 && ph == 0) __builtin_amdgcn_s_setprio(3);
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
    tr_read(0, 0); tr_rows(0); tr_read(0, 1); tr_rows(1);
    tr_cols(0, 0, 0); tr_cols(0, 0, 1); tr_cols(0, 1, 0); tr_cols(0, 1, 1);     // chunk 0 -> V[0]
  } else {
    wait_copies_keep_youngest_raw();                                // raw(1) landed; raw(2) in flight
    fix_raw(1);
  }
  lds_barrier();

  int par = 0;                                          // V/U buffer of the chunk about to be multiplied
  int slot1 = 1, slot2 = 2, slot3 = 0;                  // RAW slots of (global) chunks g+1, g+2, g+3

  // One K iteration of a transform wave: chunk kc is multiplied FIRST (while the SIMD's copy wave issues its LDS-DMA copies), then
  // chunk kc+1 (chunk 0 of the next item behind the last one) is transformed into V[par ^ 1] (while the copy wave multiplies): the
  // matrix pipe's work per chunk is a quarter of the fp32 kernel's, so what an iteration costs is each wave's own chain of LDS round
  // trips -- the two phases of a wave are kept apart (all fragment registers, then all transform registers) and staggered against
  // its partner's instead of interleaved.
  auto iter_transform = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    FISR_WB_T0
    FISR_W8B_MFMA_PHASE()
    FISR_WB_T1(0)
    tr_read(slot1, 0); tr_read2(slot1, 1);
    tr_rows(0); tr_rows2(1);
    tr_cols(par ^ 1, 0, 0); tr_cols(par ^ 1, 0, 1); tr_cols(par ^ 1, 1, 0); tr_cols(par ^ 1, 1, 1);
    FISR_WB_T1(1)
    lds_barrier();
    FISR_WB_T1(2)
    const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
    par ^= 1;
  };
  // One K iteration of a copy wave: U(kc+1) and raw(kc+3) are requested (behind an item's last chunks: the next item's first ones;
  // the copy COUNT per iteration stays fixed for the counted waits), raw(kc+2), requested an iteration ago, is finished in place
  // (relu, padding zeros), then chunk kc is multiplied.
  auto iter_copy = [&](auto first_tag, int kc) {
    constexpr bool FIRST = decltype(first_tag)::value;
    FISR_WB_T0
    if (kc == nch - 3 && has_next) raw_geom(nxt);
    if (kc == nch - 2 && has_next) fix_geom(nxt);
    if (kc == nch - 1 && has_next) u_base = (const char*)p.wpk + (size_t)nxt.nblk * W_SLAB;
    copy_u(kc + 1 < nch ? kc + 1 : (has_next ? 0 : kc), par ^ 1);
    copy_raw(kc + 3 < nch ? kc + 3 : (has_next ? kc + 3 - nch : nch - 1), slot3);
    // raw(g+2) is older than this iteration's 8 + 3 (2) copies
    if (cw < 3) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    relu_read(slot2);
    relu_write(slot2);
    zero_padding(slot2);
    FISR_WB_T1(0)
    FISR_W8B_MFMA_PHASE()
    FISR_WB_T1(1)
    wait_copies_keep_youngest_raw();      // U(g+1) landed too; raw(g+3) stays in flight
    lds_barrier();
    FISR_WB_T1(2)
    const int s_ = slot1; slot1 = slot2; slot2 = slot3; slot3 = s_;
    par ^= 1;
  };

  typedef float f2 __attribute__((ext_vector_type(2)));
#define FISR_W8_PAIR(Q, R) (f2{acc[Q][R], acc[Q][(R) + 1]})
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
    auto load_res = [&](int row, uint4 (&rres)[2][4]) {      // rres[output column j][pixel k of the quad]
      const Geo g = geometry();
      const unsigned img_bytes = (unsigned)(p.H * p.W) * (unsigned)p.Cout * 4u;
      const unsigned nrec = p.res != nullptr ? img_bytes : 0u;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)p.res + (size_t)cur.nb * p.H * p.W * p.Cout * 4), 0, nrec, 0x00020000);
      const int oy = cur.y0 + 2 * g.ty + row;
      const bool row_ok = g.c_ok & (oy < p.H);
      const unsigned rowoff = ((unsigned)(oy * p.W) * (unsigned)p.Cout + (unsigned)(g.c0 + 4 * g.lq)) * 4u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = cur.x0 + 2 * (g.txq + k) + j;
          const unsigned xo = rowoff + (unsigned)x * (unsigned)p.Cout * 4u;
          const unsigned off = (row_ok & (x < p.W)) ? xo : nrec;
          rres[j][k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    };

    // ---- K loop ----
    if (ph == 0) {
      iter_transform(first_t{});
      for (int kc = 1; kc < nch; ++kc) iter_transform(rest_t{});
    } else {
      iter_copy(first_t{}, 0);
      for (int kc = 1; kc < nch; ++kc) iter_copy(rest_t{}, kc);
    }
    uint4 rres[2][4];
    if constexpr (HAS_RES) load_res(ph, rres);
    __builtin_amdgcn_sched_barrier(0);
#ifdef FISR_WB_TRACE
    tph[3] += nch;
    const unsigned long long t_ep0 = __builtin_readcyclecounter();
#endif

    // ---- epilogue: Y = A^T M A, + bias, + residual, relu, store -- conv3x3_wino8p.h, unchanged ----
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
        if constexpr (HAS_RES) quad_transpose(rres[j], lane);
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
    if (ph == 0) {
      const int cq_shift = p.d2s_shift;
      const unsigned sub = (unsigned)c0 >> cq_shift;
      const unsigned sA = p.d2s ? (unsigned)(4 * p.W) << cq_shift : (unsigned)p.W * (unsigned)p.Cout;
      const unsigned sB = p.d2s ? 2u << cq_shift : (unsigned)p.Cout;
      const unsigned vC = (p.d2s ? ((((sub >> 1) * 2u * (unsigned)p.W + (sub & 1u)) << cq_shift) + ((unsigned)c0 & ((1u << cq_shift) - 1u)))
                                 : (unsigned)c0) + 4u * (unsigned)geo.lq;
      const unsigned out_bytes = p.d2s ? ((unsigned)(4 * p.H * p.W) << cq_shift) * 4u : (unsigned)(p.H * p.W) * (unsigned)p.Cout * 4u;
      const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
          (char*)p.out + (p.d2s ? ((size_t)cur.nb * 2 * p.H * 2 * p.W << cq_shift) * 4 : (size_t)cur.nb * p.H * p.W * p.Cout * 4), 0, out_bytes, 0x00020000);
      const float relu_lo = __builtin_bit_cast(float, p.relu_out ? 0u : 0xff800000u);
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
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            asm("v_max_f32 %0, %1, %0" : "+v"(o0[e]) : "s"(relu_lo));
            asm("v_max_f32 %0, %1, %0" : "+v"(o1[e]) : "s"(relu_lo));
          }
          rec[0][k] = __builtin_bit_cast(uint4, o0);
          rec[1][k] = __builtin_bit_cast(uint4, o1);
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          quad_transpose(rec[row], lane);
          const int oy = cur.y0 + 2 * ty + row;
          const bool row_ok = c_ok & (oy < p.H);
          const unsigned rowoff = ((unsigned)oy * sA + vC) * 4u;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = cur.x0 + 2 * (txq + k) + j;
            const unsigned xo = rowoff + (unsigned)x * sB * 4u;
            const unsigned off = (row_ok & (x < p.W)) ? xo : out_bytes;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, rec[row][k]), os, off, 0, 2);   // aux 2: nontemporal
          }
        }
      }
    }
#ifdef FISR_WB_TRACE
    tph[4] += __builtin_readcyclecounter() - t_ep0;
#endif
    if (!has_next) break;
    lds_barrier();                               // the handed-over rows are read before the next item overwrites V / U
    b_cur = b_nxt; cur = nxt;
    b_nxt = b_cur + gridDim.x;
    has_next = valid(b_nxt);
    nxt = has_next ? item_of(b_nxt) : cur;
  }
#undef FISR_W8_PAIR
#undef FISR_W8_COL
#undef FISR_W8B_MMA2
#undef FISR_W8B_MFMA_PHASE
#ifdef FISR_WB_TRACE
  if (p.trace && lane == 0) {
    tph[5] = __builtin_readcyclecounter() - t_life0;
    unsigned long long* tr = p.trace + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 6; ++i) tr[i] = tph[i];
  }
#endif
}

// ---- host side: weight slabs and launch ----

// U = G g G^T per (ci, co) in double -> Uh = bf16(U), Ul = bf16(U - Uh) (round-to-nearest-even), stored as the kernel's LDS image
// [Cin/8][Cout/64][position 16][row 64][32-byte record = {8 x Uh | 8 x Ul} of the chunk's channels], the two 16-byte halves swapped
// when bit 3 of the row is set; rows in the MFMA row order of pack_weights (fisr_api.hip): the layout of conv3x3_wino8p.h's slabs.
inline uint16_t winob_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline void pack_weights_winob(const float* w, int ci, int co, int cin_pad, std::vector<char>& wp) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nb = (co + W_BN - 1) / W_BN, nch = cin_pad / W_CH;
  wp.assign((size_t)nch * nb * W_SLAB, 0);
  for (int c = 0; c < ci; ++c)
    for (int n = 0; n < co; ++n) {
      double g[3][3], t[4][3], u[4][4];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)(a * 3 + b) * ci + c) * co + n];
      for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) u[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      const int kc = c / W_CH, cc = c % W_CH;
      const int blk = n / W_BN, nl = n % W_BN;
      const int wi = nl & 31, wk = wi >> 4, wr = wi & 15;
      const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
      char* slab = wp.data() + ((size_t)kc * nb + blk) * W_SLAB;
      for (int pos = 0; pos < 16; ++pos) {
        char* rec = slab + ((size_t)pos * 64 + row) * W_REC;
        const int sw = ((row >> 3) & 1) * 16;
        const uint16_t hi = winob_bf16((float)u[pos >> 2][pos & 3]);
        const uint32_t hb = (uint32_t)hi << 16;
        float hf;
        memcpy(&hf, &hb, 4);
        const uint16_t lo = winob_bf16((float)(u[pos >> 2][pos & 3] - (double)hf));
        reinterpret_cast<uint16_t*>(rec + (0 ^ sw))[cc] = hi;
        reinterpret_cast<uint16_t*>(rec + (16 ^ sw))[cc] = lo;
      }
    }
}

// What the kernel takes: dense NHWC fp32 tensors (no channel ranges, no dilation, no leaky relu), whole 8-channel chunks, at least
// four of them (the copy stream never runs more than one work item ahead), no fused pooling / bilinear.
inline bool winob_takes(const ConvArgs& a) {
  return a.in0_cs == a.C0 && (a.C1 == 0 || a.in1_cs == a.C1) && a.rec_cs == a.Cout && a.rec_co == 0 && a.slope == 0.f && a.dil == 1 &&
         !a.pool_out && !a.ups && (a.C0 + a.C1) / W_CH >= 4 && a.C0 % W_CH == 0 && a.C1 % W_CH == 0;
}
inline hipError_t launch_conv_winob(const ConvArgs& a, hipStream_t st) {
  static bool attr_done[64] = {};
  static int n_cu[64] = {};
  constexpr size_t lds = wino_lds_bytes();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    for (const void* k : {reinterpret_cast<const void*>(conv3x3_wino8b_kernel<false, false>), reinterpret_cast<const void*>(conv3x3_wino8b_kernel<false, true>),
                          reinterpret_cast<const void*>(conv3x3_wino8b_kernel<true, false>), reinterpret_cast<const void*>(conv3x3_wino8b_kernel<true, true>)}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    attr_done[dev] = true;
  }
  // (the epilogue addresses one output / residual image with 32-bit byte offsets)
  if (!winob_takes(a) || (double)a.H * a.W * a.Cout * 4.0 >= 4294967296.0 - 64.0) return hipErrorInvalidValue;
  const int tiles = ((a.W + TILE_W - 1) / TILE_W) * ((a.H + TILE_H - 1) / TILE_H) * a.N;
  const int items = tiles * (a.CoutPad / W_BN);
  const int grid = std::min(items, std::max(8, n_cu[dev] & ~7));      // one workgroup per CU, a multiple of 8 (conv3x3_wino8p.h)
  const bool res = a.res != nullptr;
  if (a.relu_in && res) hipLaunchKernelGGL((conv3x3_wino8b_kernel<true, true>), dim3(grid), dim3(512), lds, st, a, items);
  else if (a.relu_in) hipLaunchKernelGGL((conv3x3_wino8b_kernel<true, false>), dim3(grid), dim3(512), lds, st, a, items);
  else if (res) hipLaunchKernelGGL((conv3x3_wino8b_kernel<false, true>), dim3(grid), dim3(512), lds, st, a, items);
  else hipLaunchKernelGGL((conv3x3_wino8b_kernel<false, false>), dim3(grid), dim3(512), lds, st, a, items);
  return hipGetLastError();
}

}  // namespace fisr
