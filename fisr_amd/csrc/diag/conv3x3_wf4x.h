// conv3x3_wf4.h's F(4x4, 3x3) Winograd convolution with ONE wave per SIMD (round 4) -- a MEASURED DEAD END, kept as a diagnostics
// kernel (scripts/probes/wf4_bench.hip, WF4X=1) because the measurement is the result.
//
// Round 3's review asked for the re-tiling that breaks the LDS bound of conv3x3_wf4.h.  This is the variant that halves the weight
// fragment reads without any exchange between waves: 4 waves per workgroup, one per SIMD, each with the SIMD's whole register file
// (256 VGPRs + 256 AGPRs), a wave owning 16 channels x BOTH tile halves x 36 positions = 72 accumulators (a weight fragment is read
// once and feeds 8 MFMAs: 27 instead of 36 ds_read_b128 per 72 MFMAs), all four waves running the same stream (their quarter of the
// weight copies, raw copies and relu pass, one half-transform each), every non-MFMA instruction hand-placed in a slot behind one
// MFMA.  Same operator, tensors, weight slabs, LDS image and copy scheme as conv3x3_wf4.h, bit-identical results (self-check of the
// probe, 14 shapes).  Result on the MI355X (scripts/probes, 12 tiles, random dense data; cycles per 4-channel chunk, MFMAs 2304):
//     64->64 @544x992  4.3-4.6k (8 waves: 3.7-3.8k)   128->128 4.2k (3.5k)   256->256 4.1k (3.4k)   512->256 3.8k (3.4k)
// 8-13 % SLOWER, no spills notwithstanding.  Why (scripts/probes/mfma_queue_probe.hip): a wave that has issued a
// v_mfma_f32_16x16x4_f32 issues NOTHING else -- no s_nop, no SALU, no LDS-DMA -- until that MFMA has left the pipe: G MFMAs followed
// by F idle cycles take exactly 32 G + F cycles for every G = 1..8, one LDS-DMA copy behind 1 / 2 / 4 / 8 MFMAs costs its full 24-32
// cycles.  The fp32 MFMA runs at the vector rate on the vector lanes and holds its wave's issue slot for all 8 passes (the 16-bit
// MFMAs do not: a lone wave hides ~5 issue slots per MFMA there).  So a lone wave's stream is the SUM of its MFMAs and everything
// else (4.0-4.6k = 2304 + ~230 other instructions), and the only thing that overlaps with an fp32 MFMA is ANOTHER wave's
// non-vector instructions: two waves per SIMD are structural for this pipe, 256 registers per wave the budget, and 36 accumulators of
// 16 x 16 the most a wave can hold.  What else the work taught (all in DESIGN.md 3.1c):
//   * the compiler's MFMA selection is one form per function: with 288 accumulator registers it kept 32 of them in VGPRs and
//     shuttled them through a scratch AGPR tuple around every MFMA (copy, s_nop 9, copy) -- hence inline-asm MFMAs with "a" / "v"
//     operands chosen by hand (60 accumulators in AGPRs, 12 in VGPRs);
//   * the hazard recogniser does not see inline-asm MFMAs: a compiler-made v_mov into an accumulator two instructions ahead of the
//     asm MFMA that adds to it is read stale ("VALU write -> MFMA SrcC"), so nothing is zeroed or copied into an accumulator next
//     to the MFMAs (the second chunk of an item opens the last quad with C = 0 instead), and the epilogue's first reads are tied to
//     a wait behind the last MFMA through empty asms.
#pragma once
#include "conv3x3_wf4.h"

namespace fisr {

#ifndef FISR_F4X_TRACE
#define FISR_F4X_TRACE 0
#endif
// FISR_F4XABL (scripts/probes/wf4_bench.hip, WRONG results): 1 no weight copies, 4 no input transform, 16 no MFMAs
#ifndef FISR_F4XABL
#define FISR_F4XABL 0
#endif
#define FISR_F4X_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_F4X_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen lds\n\t"
#define FISR_F4X_END(KEEP)          "s_mov_b32 m0, %[" #KEEP "]"

template <bool RELU_IN, bool HAS_RES, bool POOL = false, bool UPS = false>
__global__ __launch_bounds__(256) void conv3x3_wf4x_kernel(const ConvArgs p, const int n_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sU = smem;
  char* const sV = smem + 2 * F4_U_BYTES;
  char* const sR = smem + 2 * F4_U_BYTES + 2 * F4_V_BYTES;
  char* const sL = smem + wf4_lds_bytes();         // (UPS only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..3: channel quarter, copy quarter, (tile half, row half) of the transform

  // ---- persistent work-item walk: conv3x3_wf4.h ----
  const int tiles_x = (p.W + F4_TW - 1) / F4_TW, tiles_y = (p.H + F4_TH - 1) / F4_TH;
  const int nblocks = p.CoutPad / F4_BN;
  struct Item { int x0, y0, nb, nblk; };
  auto item_of = [&](int b) __attribute__((always_inline)) {
    const int q = n_items >> 3, r = n_items & 7;
    const int xcd = b & 7, loc = b >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int t = v / nblocks;
    Item it;
    it.nblk = v - t * nblocks;
    const int tx = t % tiles_x; t /= tiles_x;
    it.x0 = tx * F4_TW;
    it.y0 = (t % tiles_y) * F4_TH;
    it.nb = t / tiles_y;
    return it;
  };
  int b_cur = blockIdx.x;
  Item cur = item_of(b_cur);
  bool has_next = b_cur + (int)gridDim.x < n_items;
  Item nxt = has_next ? item_of(b_cur + gridDim.x) : cur;
  const int nch0 = p.C0 / F4_CH, nch = (p.C0 + p.C1) / F4_CH;       // (the launcher guarantees nch >= 4)

  unsigned long long t_start = 0, t_first = 0, t_main = 0, t_end1 = 0, t_real = 0;
  if (FISR_F4X_TRACE && p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  // =========================== copies (LDS-DMA, 1 KB per wave instruction) ===========================
  // weight copy c (0..35) of a slab = [position quad 9][channel quarter 4] 1-KB pieces: wave w issues the pieces w + 4 j, j = 0..8
  // -- the nine pieces of its OWN channel quarter, the only ones it reads.
  // raw copy c (0..19) of a raw pair moves halo slots 32 c .. 32 c + 31 (two lanes per slot); wave w issues pieces w, w + 4, .., w + 16.
  const int cw = wave;
  constexpr unsigned OOB = 0x80000000u;
  const size_t img_px = (size_t)p.H * p.W;
  int rpix[5] = {-1, -1, -1, -1, -1};              // pixel index of the lane's halo slot per piece, or -1 (outside the image)
  unsigned rsub = 0;                               // bit j: which chunk of the pair the lane fetches in piece j
  __amdgpu_buffer_rsrc_t rs0, rs1;
  auto raw_geom = [&](const Item& it) __attribute__((always_inline)) {
    rsub = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int s = 32 * (cw + 4 * q) + (l >> 1);
      const int py = s / F4_HW, r = s - py * F4_HW;
      const int px = r < 9 ? 4 * r : r < 18 ? 4 * (r - 9) + 1 : r < 26 ? 4 * (r - 18) + 2 : 4 * (r - 26) + 3;
      const int gy = it.y0 - 1 + py, gx = it.x0 - 1 + px;
      const bool ok = s < F4_HALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      rpix[q] = ok ? gy * p.W + gx : -1;
      rsub |= (unsigned)((l & 1) ^ ((py >> 2) & 1)) << q;
    }
    rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.in0 + (size_t)it.nb * img_px * p.C0), 0, (unsigned)(img_px * p.C0 * 4), 0x00020000);
    rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? (const float*)p.in1 + (size_t)it.nb * img_px * p.C1 : (const float*)p.in0), 0,
                                            (unsigned)(img_px * (p.in1 ? p.C1 : p.C0) * 4), 0x00020000);
  };
  unsigned lro[2] = {OOB, OOB};
  auto ups_geom = [&](const Item& it) __attribute__((always_inline)) {
    const int hl = p.H >> 1, wl = p.W >> 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int wp = j == 0 ? cw : 4 + cw;
      const int plane = wp >= 3 ? 1 : 0;
      const int s = (wp - 3 * plane) * 64 + l;
      const int ly = s / F4_L_COLS, lx = s - ly * F4_L_COLS;
      const int sy = min(max((it.y0 >> 1) - 1 + ly, 0), hl - 1), sx = min(max((it.x0 >> 1) - 1 + lx, 0), wl - 1);
      lro[j] = s < F4_L_ROWS * F4_L_COLS ? (unsigned)(sy * wl + sx) * (unsigned)(p.C0 * 4) + (unsigned)plane * 16u : OOB;
    }
    rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.in0 + (size_t)it.nb * ((size_t)hl * wl) * p.C0), 0, 0x7fffffffu, 0x00020000);
  };
  if constexpr (UPS) ups_geom(cur); else raw_geom(cur);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (unsigned)((size_t)nch * nblocks * F4_U_BYTES), 0x00020000);
  const unsigned raw_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sR + (unsigned)cw * 1024u;
  const unsigned u_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sU;
  const unsigned u_voff = (unsigned)lane * 16u;
  const unsigned u_rot = ((blockIdx.x >> 3) * 7u) % 9u;        // (the workgroups of an XCD walk a slab's position quads from different starting points)

#define FISR_F4X_DMA1(RS, VOFF, SOFF, LDS)                                                                      \
  do {                                                                                                          \
    unsigned keep_;                                                                                             \
    asm volatile(FISR_F4X_BEGIN(keep, lds) FISR_F4X_COPY(o, rs, so) FISR_F4X_END(keep)                          \
                 : [keep] "=&s"(keep_) : [rs] "s"(RS), [lds] "s"(LDS), [o] "v"(VOFF), [so] "s"(SOFF) : "memory", "scc"); \
  } while (0)
  // weight copy j (0..8) of chunk kc of N block nblk into U[buf]
  auto copy_u1 = [&](int nblk, int kc, int buf, int j) __attribute__((always_inline)) {
    if (FISR_F4XABL & 1) return;
    unsigned qd = (unsigned)j + u_rot;
    qd = qd >= 9u ? qd - 9u : qd;
    const unsigned c = qd * 4u + (unsigned)cw;
    const unsigned so = (unsigned)(((size_t)kc * nblocks + nblk) * F4_U_BYTES) + c * 1024u;
    const unsigned lds = u_lds0 + (unsigned)buf * (unsigned)F4_U_BYTES + c * 1024u;
    FISR_F4X_DMA1(rsw, u_voff, so, lds);
  };
  unsigned ro[5] = {OOB, OOB, OOB, OOB, OOB};
  bool ro_first = true;
  auto raw_offsets = [&](bool first) __attribute__((always_inline)) {
    const unsigned csb = (unsigned)(first ? p.C0 : p.C1) * 4u;
#pragma unroll
    for (int j = 0; j < 5; ++j) ro[j] = rpix[j] < 0 ? OOB : (unsigned)rpix[j] * csb + ((rsub >> j) & 1u) * 16u;
    ro_first = first;
  };
  auto copy_pair1 = [&](int pc, int pb, int j) __attribute__((always_inline)) {
    const bool first = 2 * pc < nch0;
    const unsigned so = (unsigned)(first ? pc : pc - (nch0 >> 1)) * 32u;
    const unsigned lds = raw_lds0 + (unsigned)pb * (unsigned)(2 * F4_RAW_BYTES) + (unsigned)j * 4096u;
    const unsigned o = j == 0 ? ro[0] : j == 1 ? ro[1] : j == 2 ? ro[2] : j == 3 ? ro[3] : ro[4];
    if (first) FISR_F4X_DMA1(rs0, o, so, lds); else FISR_F4X_DMA1(rs1, o, so, lds);
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // relu-on-load, once per element in LDS, by the wave that requested the copy: piece j (0..4) of pair buffer pb
  f32x4 rl[3];
  auto relu_read1 = [&](int pb, int j, int slot) __attribute__((always_inline)) {
    rl[slot] = *reinterpret_cast<const f32x4*>(sR + pb * (2 * F4_RAW_BYTES) + cw * 1024 + lane * 16 + j * 4096);
  };
  auto relu_write1 = [&](int pb, int j, int slot) __attribute__((always_inline)) {
    f32x4 f = rl[slot];
    asm("v_max_f32 %0, 0, %0" : "+v"(f.x)); asm("v_max_f32 %0, 0, %0" : "+v"(f.y));
    asm("v_max_f32 %0, 0, %0" : "+v"(f.z)); asm("v_max_f32 %0, 0, %0" : "+v"(f.w));
    *reinterpret_cast<f32x4*>(sR + pb * (2 * F4_RAW_BYTES) + cw * 1024 + lane * 16 + j * 4096) = f;
  };

  // UPS: staged piece j (0; 1: waves 0-1 only) of pair pc into staging buffer lb
  const unsigned l_lds0 = (unsigned)(size_t)(wf4_lds_ptr_t)sL;
  auto copy_l1 = [&](int pc, int lb, int j) __attribute__((always_inline)) {
    const unsigned so = (unsigned)pc * 32u;
    const unsigned lds = l_lds0 + (unsigned)lb * (unsigned)F4_L_BYTES + (unsigned)(j == 0 ? cw : 4 + cw) * 1024u;
    const unsigned o = j == 0 ? lro[0] : lro[1];
    FISR_F4X_DMA1(rs0, o, so, lds);
  };
  // UPS: quad q = 64 cw + lane of the 9 x 17 quads of a halo tile (waves 0-2)
  int e_la = 0, e_r0 = 0, e_r1 = 0;
  if constexpr (UPS) {
    const int q = min(cw * 64 + lane, 152), qy = q / 17, qx = q - qy * 17;
    e_la = (qy * F4_L_COLS + qx) * 16;
    const int f16 = ((qy >> 1) & 1) * 16;
    e_r0 = (2 * qy * F4_HW + ((qx & 1) ? 18 : 0) + (qx >> 1)) * 32 + f16;
    e_r1 = (2 * qy * F4_HW + ((qx & 1) ? 26 : 9) + (qx >> 1)) * 32 + f16;
  }
  f32x4 e_tl, e_tr, e_bl, e_br;
  auto ex_read = [&](int lb, int c) __attribute__((always_inline)) {
    const char* b = sL + lb * F4_L_BYTES + c * F4_L_PLANE + e_la;
    e_tl = *reinterpret_cast<const f32x4*>(b);
    e_tr = *reinterpret_cast<const f32x4*>(b + 16);
    e_bl = *reinterpret_cast<const f32x4*>(b + F4_L_COLS * 16);
    e_br = *reinterpret_cast<const f32x4*>(b + F4_L_COLS * 16 + 16);
  };
  auto ex_write = [&](int pb, int c, int ey0, int ex0) __attribute__((always_inline)) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    auto fma2 = [](f2_t a, f2_t c2) { f2_t r; asm("v_pk_fma_f32 %0, %1, 0.5, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c2)); return r; };
    auto sub2 = [](f2_t a, f2_t b2) { f2_t r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b2)); return r; };
    f32x4 oD, oC, oB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f2_t tl = {e_tl[2 * h], e_tl[2 * h + 1]}, tr = {e_tr[2 * h], e_tr[2 * h + 1]};
      const f2_t bl = {e_bl[2 * h], e_bl[2 * h + 1]}, br = {e_br[2 * h], e_br[2 * h + 1]};
      const f2_t A = fma2(sub2(tr, tl), tl);
      const f2_t B = fma2(sub2(br, bl), bl);
      const f2_t C = fma2(sub2(br, tr), tr);
      const f2_t D = fma2(sub2(B, A), A);
      oD[2 * h] = D.x; oD[2 * h + 1] = D.y; oC[2 * h] = C.x; oC[2 * h + 1] = C.y; oB[2 * h] = B.x; oB[2 * h + 1] = B.y;
    }
    f32x4 oE = e_br;
    const bool interior = ey0 > 0 && ex0 > 0 && ey0 + F4_TH + 1 <= p.H && ex0 + F4_TW + 1 <= p.W;
    if (!interior) {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int q = min(cw * 64 + l, 152), qy = q / 17, qx = q - qy * 17;
      const int gy = ey0 - 1 + 2 * qy, gx = ex0 - 1 + 2 * qx;
      const bool ya = gy >= 0 && gy < p.H, yb = gy + 1 < p.H, xa = gx >= 0 && gx < p.W, xb = gx + 1 < p.W;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      oD = ya && xa ? oD : z; oC = ya && xb ? oC : z; oB = yb && xa ? oB : z; oE = yb && xb ? oE : z;
    }
    char* r = sR + pb * (2 * F4_RAW_BYTES);
    const int a0 = e_r0 ^ (c * 16), a1 = e_r1 ^ (c * 16);
    *reinterpret_cast<f32x4*>(r + a0) = oD;
    *reinterpret_cast<f32x4*>(r + a1) = oC;
    *reinterpret_cast<f32x4*>(r + a0 + F4_HW * 32) = oB;
    *reinterpret_cast<f32x4*>(r + a1 + F4_HW * 32) = oE;
  };

  // =========================== input transform (tile half wave & 1, rows 3 (wave >> 1) .. of B^T d B): conv3x3_wf4.h ===========================
  const int t_ch = lane & 3, t_t8 = (lane >> 2) & 7;
  const int t_tx = (lane >> 5) * 4 + (t_t8 & 3), t_tyl = t_t8 >> 2;
  const int t_ty = 2 * (wave & 1) + t_tyl, t_t16 = t_tyl * 8 + t_tx;
  const int t_ra = ((4 * t_ty) * F4_HW + t_tx) * 32 + (t_ty & 1) * 16 + t_ch * 4;
  const int t_rb = t_ra ^ 16;
  const int t_voff = (wave & 1) * 1024 + (t_ch * 16 + (t_t16 ^ (t_ch << 1))) * 16;
  const f32x2 K8 = {8.f, 8.f}, K4 = {4.f, 4.f}, KM4 = {-4.f, -4.f}, KM5 = {-5.f, -5.f}, K2 = {2.f, 2.f}, K41 = {-4.f, -1.f}, K12 = {1.f, 2.f};
  auto pk_fma = [](f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; };
  auto pk_fnma = [](f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; };
  auto pk_add = [](f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
  auto pk_sub = [](f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; };
  f32x2 dp[6][3];                                  // the 6 x 6 patch of this lane's (tile, channel) as column pairs
  f32x2 zp[3][3];                                  // this wave's three rows: vertical pass, then horizontal pass in place
  auto tr_read_row = [&](int pb, int sub, int r) __attribute__((always_inline)) {       // patch row r of chunk `sub` of the pair in buffer pb
    const char* r03 = sR + pb * (2 * F4_RAW_BYTES) + (sub ? t_rb : t_ra);
    const char* r45 = sR + pb * (2 * F4_RAW_BYTES) + (sub ? t_ra : t_rb);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int cb = (j & 3) == 0 ? 0 : (j & 3) == 1 ? 9 : (j & 3) == 2 ? 18 : 26;     // column 4 tx + j sits at cb + tx + j / 4
      dp[r][j >> 1][j & 1] = *reinterpret_cast<const float*>((r < 4 ? r03 : r45) + (r * F4_HW + cb + (j >> 2)) * 32);
    }
  };
  auto tr_col = [&](auto rh_tag, int c) __attribute__((always_inline)) {
    constexpr int RH = decltype(rh_tag)::value;
    if constexpr (RH == 0) {
      const f32x2 a = pk_fma(dp[2][c], KM4, dp[4][c]), b = pk_fma(dp[1][c], KM4, dp[3][c]);
      zp[0][c] = pk_fma(dp[0][c], K4, pk_fma(dp[2][c], KM5, dp[4][c]));
      zp[1][c] = pk_add(a, b);
      zp[2][c] = pk_sub(a, b);
    } else {
      const f32x2 cc = pk_sub(dp[4][c], dp[2][c]), e = pk_sub(dp[3][c], dp[1][c]);
      zp[0][c] = pk_fma(e, K2, cc);
      zp[1][c] = pk_fnma(e, K2, cc);
      zp[2][c] = pk_fma(dp[1][c], K4, pk_fma(dp[3][c], KM5, dp[5][c]));
    }
  };
  auto tr_row = [&](int i) __attribute__((always_inline)) {                       // -> (t0, t5), (t1, t3), (t2, t4)
    const f32x2 p01 = zp[i][0], p23 = zp[i][1], p45 = zp[i][2];
    f32x2 ac, be, t05, t13, t24;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(ac) : "v"(p23), "s"(K41), "v"(p45));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(be) : "v"(p01), "s"(K41), "v"(p23));
    t13 = pk_fma(be, K12, ac);
    t24 = pk_fnma(be, K12, ac);
    t05 = pk_fma(p01, K4, pk_fma(p23, KM5, p45));
    zp[i][0] = t05; zp[i][1] = t13; zp[i][2] = t24;
  };
  auto quad_of = [](f32x2 a, f32x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); };
  auto tr_write = [&](auto rh_tag, int vbuf, int i) __attribute__((always_inline)) {     // what row i (0..2) of this wave completes
    constexpr int RH = decltype(rh_tag)::value;
    char* vb = sV + vbuf * F4_V_BYTES + t_voff;
    if constexpr (RH == 0) {
      if (i == 0) *reinterpret_cast<f32x4*>(vb) = quad_of(zp[0][0], zp[0][1]);
      if (i == 1) {
        *reinterpret_cast<f32x4*>(vb + 2048) = quad_of(zp[0][2], zp[1][0]);
        *reinterpret_cast<f32x4*>(vb + 2 * 2048) = quad_of(zp[1][1], zp[1][2]);
      }
      if (i == 2) {
        *reinterpret_cast<f32x4*>(vb + 3 * 2048) = quad_of(zp[2][0], zp[2][1]);
        *reinterpret_cast<f32x2*>(vb + 4 * 2048) = zp[2][2];
      }
    } else {
      if (i == 0) {
        *reinterpret_cast<f32x2*>(vb + 4 * 2048 + 8) = zp[0][0];
        *reinterpret_cast<f32x4*>(vb + 5 * 2048) = quad_of(zp[0][1], zp[0][2]);
      }
      if (i == 1) *reinterpret_cast<f32x4*>(vb + 6 * 2048) = quad_of(zp[1][0], zp[1][1]);
      if (i == 2) {
        *reinterpret_cast<f32x4*>(vb + 7 * 2048) = quad_of(zp[1][2], zp[2][0]);
        *reinterpret_cast<f32x4*>(vb + 8 * 2048) = quad_of(zp[2][1], zp[2][2]);
      }
    }
  };
  typedef std::integral_constant<int, 0> rh0_t;
  typedef std::integral_constant<int, 1> rh1_t;
  auto transform = [&](auto rh_tag, int pb, int sub, int vbuf) __attribute__((always_inline)) {       // (prologue: the whole transform at once)
#pragma unroll
    for (int r = 0; r < 6; ++r) tr_read_row(pb, sub, r);
#pragma unroll
    for (int c = 0; c < 3; ++c) tr_col(rh_tag, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) { tr_row(i); tr_write(rh_tag, vbuf, i); }
  };

  // =========================== MFMA side: channel quarter cq = wave, both tile halves ===========================
  const int cq = wave;
  const char* const fu = sU + cq * 1024 + lane * 16;
  const char* const fv = sV + ((lane & 0x30) | ((lane & 15) ^ ((lane >> 4) << 1))) * 16;      // tile half th: + 1024 th
  f32x4 acc[72];                                   // [tile half][slot of the position]
  // ring of three position quads: the weights' fragment (all 16 channels x 4 ci of four positions) feeds both tile halves;
  // slot 2 holds the pending last quad of a chunk across the barrier (conv3x3_wf4.h's skew)
  f32x4 ra[3], rb[2][3];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bias4 = zero4;
  static_assert(wf4_slot(1, 1) == 8, "the bias rides in accumulator 8 (quad 2, element 0)");
  // The MFMAs are inline asm: 72 accumulators are 288 registers, 256 of which can be AGPRs -- and the compiler's MFMA selection is
  // one form per function (AGPR destination), so it shuttled the accumulators that live in VGPRs through a scratch AGPR tuple around
  // every MFMA (copy in, s_nop 9, copy out: the pipe drained 12 times per chunk).  By hand: slots 0..27 (and 28..31 of tile half 0)
  // are "a" operands, the last quad (and slots 28..31 of tile half 1) "v" operands -- 240 AGPRs + 48 VGPRs.  What the hazard
  // recogniser no longer sees is kept by construction: an accumulator is touched once per 72 MFMAs (no back-to-back SrcC
  // dependency), fragments come from ds_reads (s_waitcnt, which the compiler still inserts for asm operands), and the epilogue
  // waits 32 states behind the last MFMA before it reads.
#define FISR_F4X_ACC_V(TH, S) ((S) >= 32 || ((TH) == 1 && (S) >= 28))
#define FISR_F4X_MMA1(TH, Q, E, A, B)                                                                              \
  if (!(FISR_F4XABL & 16)) {                                                                                       \
    if (FISR_F4X_ACC_V(TH, 4 * (Q) + (E)))                                                                          \
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[36 * (TH) + 4 * (Q) + (E)]) : "v"((A)[E]), "v"((B)[E]));       \
    else                                                                                                           \
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[36 * (TH) + 4 * (Q) + (E)]) : "v"((A)[E]), "v"((B)[E]));       \
  }
  // first chunk of an item: C = 0 (nothing is zeroed by hand) -- except accumulator (1,1), which was set to the bias
#define FISR_F4X_MMA1Z(TH, Q, E, A, B)                                                                             \
  if (!(FISR_F4XABL & 16)) {                                                                                       \
    if (FISR_F4X_ACC_V(TH, 4 * (Q) + (E)))                                                                          \
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[36 * (TH) + 4 * (Q) + (E)]) : "v"((A)[E]), "v"((B)[E]));        \
    else                                                                                                           \
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[36 * (TH) + 4 * (Q) + (E)]) : "v"((A)[E]), "v"((B)[E]));         \
  }

  // ---- prologue of the workgroup's FIRST item: raw pair 0, U(0), raw pair 1 requested; chunk 0 -> V[0] ----
  if constexpr (UPS) {           // staged pairs 0 and 1, U(0); everything landed before the barrier (the quads read other waves' pieces)
    copy_l1(0, 0, 0);
    if (cw < 2) copy_l1(0, 0, 1);
#pragma unroll
    for (int j = 0; j < 9; ++j) copy_u1(cur.nblk, 0, 0, j);
    copy_l1(1, 1, 0);
    if (cw < 2) copy_l1(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    raw_offsets(true);
#pragma unroll
    for (int j = 0; j < 5; ++j) copy_pair1(0, 0, j);
#pragma unroll
    for (int j = 0; j < 9; ++j) copy_u1(cur.nblk, 0, 0, j);
    if (2 >= nch0) raw_offsets(false);
#pragma unroll
    for (int j = 0; j < 5; ++j) copy_pair1(1, 1, j);
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // pair 0, U(0) landed; pair 1 in flight
    if constexpr (RELU_IN) {
#pragma unroll
      for (int j = 0; j < 5; ++j) { relu_read1(0, j, 0); relu_write1(0, j, 0); }
    }
  }
  lds_barrier();
  if constexpr (UPS) {
    if (cw < 3) { ex_read(0, 0); ex_write(0, 0, cur.y0, cur.x0); ex_read(0, 1); ex_write(0, 1, cur.y0, cur.x0); }
    lds_barrier();
  }
  if (wave < 2) transform(rh0_t{}, 0, 0, 0);
  else transform(rh1_t{}, 0, 0, 0);
  lds_barrier();
  if (FISR_F4X_TRACE && p.trace) t_first = __builtin_readcyclecounter();

  // ---- K loop of one item ----
  // Iteration g (the workgroup's running chunk counter; parity = k's) multiplies chunk g and transforms chunk g+1 out of its raw pair;
  // raw pairs as in conv3x3_wf4.h (pair m+1 requested in the odd iteration 2m-1, relu'd late in 2m, read from 2m+1 on).  An
  // iteration is 9 groups of 8 MFMAs (group 0: the pending last quad of the chunk before, from registers; groups 1-8: quads 0-7),
  // and behind EVERY MFMA sits one slot of the rest of the wave's stream, fenced with sched_barrier: a lone wave issues in order,
  // so whatever is not placed between two MFMAs is paid in full.
  //   slot (g, 0), g = 1..7      fragments of quad g + 1 (three ds_read_b128)
  //   weight copies 0..8         (0,1) (1,1) (1,3) (1,5) (2,1) (3,1) (4,1) (5,1) (6,1): all in front of the raw copies -- vmcnt is one
  //                              in-order counter, "U(g+1) landed, the raw pair in flight" needs the pair to be the youngest
  //   transform of chunk g + 1   patch rows (0, 2..7); vertical pass (2, 2) (2, 4) (2, 6); rows + V writes (3..5, 2) / (3..5, 5)
  //   odd: raw pair copies       (6,3) (6,6) (7,3) (7,6) (8,3)
  //   even, relu-on-load         wait for the pair, reads (5, 3..5) (7, 3..4), max + writes (6, 3..5) (8, 3..4)
  //   even, fused bilinear       blend (5,3) (6,3) (7,3) (8,3), staged copies (6,5) (6,6)
  int par = 0;                                     // V / U buffer of the chunk about to be multiplied
  int pbt = 0;                                     // raw pair buffer that holds chunk g+1
  typedef std::integral_constant<int, 1> first_t;
  typedef std::integral_constant<int, 2> second_t;
  typedef std::integral_constant<int, 0> rest_t;
  typedef std::integral_constant<bool, true> odd_t;
  typedef std::integral_constant<bool, false> even_t;
  // (FIRST = 2: the SECOND chunk of an item, whose group 0 opens the accumulators of the last quad with C = 0.  Nothing is zeroed
  //  or copied into an accumulator by compiler-made vector instructions next to the asm MFMAs: the hazard recogniser does not see
  //  them, and a v_mov into an accumulator two instructions ahead of the MFMA that adds to it was read stale -- "VALU write ->
  //  MFMA SrcC" wants wait states; measured as two wrong channels of four in one output row of two instantiations.)
  auto k_iter = [&](auto rh_tag, auto first_tag, auto odd_tag, int k) __attribute__((always_inline)) {
    typedef decltype(rh_tag) RH;
    constexpr bool FIRST = decltype(first_tag)::value == 1;     // first chunk of an item
    constexpr bool SECOND = decltype(first_tag)::value == 2;
    constexpr bool ODD = decltype(odd_tag)::value;         // k odd
    const int buf = par;
    const char* ub = fu + buf * F4_U_BYTES;
    const char* vb = fv + buf * F4_V_BYTES;
    // U(k+1): of this item, of the next item (chunk 0), or a repeated chunk behind the last item
    const bool u_here = k + 1 < nch;
    const int ku = u_here ? k + 1 : (has_next ? 0 : nch - 1);
    const int nblk_cur = cur.nblk, nblk_nxt = nxt.nblk;
    const int u_nblk = u_here || !has_next ? nblk_cur : nblk_nxt;
    auto frag = [&](int q) __attribute__((always_inline)) {
      ra[q % 3] = *reinterpret_cast<const f32x4*>(ub + q * 4096);
      rb[0][q % 3] = *reinterpret_cast<const f32x4*>(vb + q * 2048);
      rb[1][q % 3] = *reinterpret_cast<const f32x4*>(vb + q * 2048 + 1024);
    };
    frag(0);
    frag(1);
    int pc = 0;
    int ey0 = 0, ex0 = 0;
    if constexpr (UPS && !ODD) {
      // even iterations blend the pair (k + 2) / 2 out of staging buffer pbt ^ 1 and request the pair (k + 4) / 2 into the other
      const int pp = (k + 4) >> 1, np = nch >> 1;
      pc = pp < np ? pp : (has_next ? pp - np : np - 1);
      if (k == nch - 4 && has_next) ups_geom(nxt);
      const int cy0 = cur.y0, cx0 = cur.x0, ny0 = nxt.y0, nx0 = nxt.x0;
      const bool of_next = k + 2 >= nch;
      ey0 = of_next ? ny0 : cy0; ex0 = of_next ? nx0 : cx0;
    }
    if constexpr (ODD && !UPS) {
      const int pp = (k + 3) >> 1, np = nch >> 1;
      pc = pp < np ? pp : (has_next ? pp - np : np - 1);
      const bool rfirst = 2 * pc < nch0;
      bool redo = rfirst != ro_first;
      if (k == nch - 3 && has_next) { raw_geom(nxt); redo = true; }
      if (redo) raw_offsets(rfirst);
    }
    if constexpr (FIRST) {
      acc[8] = bias4; acc[44] = bias4;             // position (1, 1): A^T has a 1 in column 1 of every row, so M[1][1] enters all 16 outputs once
      asm volatile("s_nop 4" : "+a"(acc[8]), "+a"(acc[44]));      // (the copies into the AGPRs, then wait states, then the MFMAs)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int th = m >> 2, e = m & 3;
        if (g == 0) {
          if constexpr (SECOND) { FISR_F4X_MMA1Z(th, 8, e, ra[2], rb[th][2]) }
          else if constexpr (!FIRST) { FISR_F4X_MMA1(th, 8, e, ra[2], rb[th][2]) }
        } else {
          const int q = g - 1;
          if constexpr (FIRST) {
            if (q == 2 && e == 0) { FISR_F4X_MMA1(th, q, e, ra[q % 3], rb[th][q % 3]) } else { FISR_F4X_MMA1Z(th, q, e, ra[q % 3], rb[th][q % 3]) }
          } else { FISR_F4X_MMA1(th, q, e, ra[q % 3], rb[th][q % 3]) }
        }
        // ---- the slot behind this MFMA ----
        if (m == 0 && g >= 1 && g <= 7) frag(g + 1);
        {
          const int j = (g == 0 && m == 1) ? 0 : (g == 1 && m == 1) ? 1 : (g == 1 && m == 3) ? 2 : (g == 1 && m == 5) ? 3
                      : (g >= 2 && g <= 6 && m == 1) ? g + 2 : -1;
          if (j >= 0) copy_u1(u_nblk, ku, buf ^ 1, j);
        }
        if (!(FISR_F4XABL & 4)) {
          if (g == 0 && m >= 2) tr_read_row(pbt, ODD ? 0 : 1, m - 2);
          if (g == 2 && (m == 2 || m == 4 || m == 6)) tr_col(RH{}, (m - 2) >> 1);
          if (g >= 3 && g <= 5 && m == 2) tr_row(g - 3);
          if (g >= 3 && g <= 5 && m == 5) tr_write(RH{}, buf ^ 1, g - 3);
        }
        if constexpr (UPS) {
          if constexpr (!ODD) {
            if (m == 3 && cw < 3) {
              if (g == 5) ex_read(pbt ^ 1, 0);
              if (g == 6) ex_write(pbt ^ 1, 0, ey0, ex0);
              if (g == 7) ex_read(pbt ^ 1, 1);
              if (g == 8) ex_write(pbt ^ 1, 1, ey0, ex0);
            }
            if (g == 6 && m == 5) copy_l1(pc, pbt, 0);
            if (g == 6 && m == 6 && cw < 2) copy_l1(pc, pbt, 1);
          }
        } else if constexpr (ODD) {
          if (g == 6 && m == 3) copy_pair1(pc, pbt ^ 1, 0);
          if (g == 6 && m == 6) copy_pair1(pc, pbt ^ 1, 1);
          if (g == 7 && m == 3) copy_pair1(pc, pbt ^ 1, 2);
          if (g == 7 && m == 6) copy_pair1(pc, pbt ^ 1, 3);
          if (g == 8 && m == 3) copy_pair1(pc, pbt ^ 1, 4);
        } else if constexpr (RELU_IN) {
          // the pair requested an iteration ago is older than the eight weight copies of this iteration so far
          if (g == 5 && m == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          if (g == 5 && m >= 3 && m <= 5) relu_read1(pbt ^ 1, m - 3, m - 3);
          if (g == 6 && m >= 3 && m <= 5) relu_write1(pbt ^ 1, m - 3, m - 3);
          if (g == 7 && m >= 3 && m <= 4) relu_read1(pbt ^ 1, m, m - 3);
          if (g == 8 && m >= 3 && m <= 4) relu_write1(pbt ^ 1, m, m - 3);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (UPS) {
      if constexpr (!ODD) {                                                    // U(g+1) landed; the staged pair stays in flight
        if (cw < 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // U(g+1) and the staged pair of the iteration before
    } else if constexpr (ODD) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // U(g+1) landed; the raw pair stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // U(g+1) and the raw pair of the iteration before
    lds_barrier();
    if (!ODD) pbt ^= 1;
    par ^= 1;
  };
  unsigned long long t2[6] = {0, 0, 0, 0, 0, 0};   // timeline of the workgroup's SECOND item (steady state), trace runs only
  int n_done = 0;
  auto k_loop = [&](auto rh_tag) __attribute__((always_inline)) {
    const bool tr2 = FISR_F4X_TRACE && p.trace && n_done == 1;
    if (tr2) t2[0] = __builtin_readcyclecounter();
    k_iter(rh_tag, first_t{}, even_t{}, 0);
    if (tr2) t2[1] = __builtin_readcyclecounter();
    k_iter(rh_tag, second_t{}, odd_t{}, 1);
    if (tr2) t2[2] = __builtin_readcyclecounter();
    for (int k = 2; k < nch; k += 2) { k_iter(rh_tag, rest_t{}, even_t{}, k); k_iter(rh_tag, rest_t{}, odd_t{}, k + 1); }
    if (tr2) t2[3] = __builtin_readcyclecounter();
  };

  for (;;) {                                       // ---- work items of this workgroup ----
    {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int c0 = cur.nblk * F4_BN + cq * 16 + 4 * (l >> 4);
      bias4 = *reinterpret_cast<const f32x4*>(p.bias + (c0 < p.Cout ? c0 : 0));
    }
    if (wave < 2) k_loop(rh0_t{});
    else k_loop(rh1_t{});
#pragma unroll
    for (int m = 0; m < 8; ++m) { FISR_F4X_MMA1((m >> 2), 8, (m & 3), ra[2], rb[m >> 2][2]) }     // the last chunk's pending quad
    // The last MFMAs' results: 8 passes + margin before anything reads them.  The hazard recogniser does not see inline-asm MFMAs,
    // and a plain `s_nop` asm orders nothing the compiler schedules around it -- the reads of these eight accumulators are tied
    // to the wait through the empty asms behind it (volatile asms keep their order; without them the first output row that needs
    // accumulator row 5 read stale registers in one instantiation).
    // (measured: MFMAs QUEUE -- eight of them issue within a few dozen cycles and complete 32 cycles apart, so the wait behind the
    //  last one has to cover the whole queue, 8 x 32 cycles + write-back; two s_nop 15 read stale rows in two instantiations)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
                 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) { asm volatile("" : "+v"(acc[32 + r])); asm volatile("" : "+v"(acc[68 + r])); }
    if (FISR_F4X_TRACE && p.trace && n_done == 0) t_main = __builtin_readcyclecounter();

    // ---- epilogue (conv3x3_wf4.h), once per tile half: Y = A^T M A in registers, lane transposition, (+ residual), relu, 16-byte stores ----
    if (FISR_F4X_TRACE && p.trace && n_done == 1) t2[4] = __builtin_readcyclecounter();
    auto epilogue = [&](auto th_tag) __attribute__((always_inline)) {
      constexpr int TH = decltype(th_tag)::value;
      int l = lane;
      asm volatile("" : "+v"(l));                  // (recomputed per item and tile half: hoisted, it would stay live across the K loops)
      const int bp_addr = (((l & 3) << 4) | (l >> 2)) << 2;       // byte address of the SOURCE lane of the transposition
      const int e_t = TH * 16 + (l >> 2);
      const int e_ty = e_t >> 3, e_tx = e_t & 7;
      const int c0 = cur.nblk * F4_BN + cq * 16 + 4 * (l & 3);
      const bool c_ok = c0 < p.Cout;
      const int cq_shift = p.d2s_shift;
      const unsigned sub = (unsigned)c0 >> cq_shift;
      const unsigned sA = p.d2s ? (unsigned)(4 * p.W) << cq_shift : (unsigned)p.W * (unsigned)p.Cout;
      const unsigned sB = p.d2s ? 2u << cq_shift : (unsigned)p.Cout;
      const unsigned vC = p.d2s ? ((((sub >> 1) * 2u * (unsigned)p.W + (sub & 1u)) << cq_shift) + ((unsigned)c0 & ((1u << cq_shift) - 1u))) : (unsigned)c0;
      const unsigned out_bytes = p.d2s ? ((unsigned)(4 * p.H * p.W) << cq_shift) * 4u : (unsigned)(p.H * p.W) * (unsigned)p.Cout * 4u;
      const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
          (char*)p.out + (p.d2s ? ((size_t)cur.nb * 2 * p.H * 2 * p.W << cq_shift) * 4 : (size_t)cur.nb * img_px * p.Cout * 4), 0, out_bytes, 0x00020000);
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      const int oy0 = cur.y0 + 4 * e_ty, ox0 = cur.x0 + 4 * e_tx;
      const unsigned vbase = c_ok ? ((unsigned)oy0 * sA + (unsigned)ox0 * sB + vC) * 4u : OOB;
      const bool interior = cur.y0 + F4_TH <= p.H && cur.x0 + F4_TW <= p.W;      // (uniform)
      unsigned off[4][4];
      if (!interior) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) off[i][j] = ((oy0 + i < p.H) & (ox0 + j < p.W)) ? vbase : OOB;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) off[i][j] = vbase;
      }
      const unsigned sA4 = sA * 4u, sB4 = sB * 4u;
#define FISR_F4X_SOFF(I, J) ((unsigned)(I) * sA4 + (unsigned)(J) * sB4)
      auto at_pk = [&](f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2& y0, f32x2& y1, f32x2& y2, f32x2& y3) __attribute__((always_inline)) {
        const f32x2 s1 = pk_add(m1, m2), d1 = pk_sub(m1, m2), s2 = pk_add(m3, m4), d2 = pk_sub(m3, m4);
        y0 = pk_add(pk_add(m0, s1), s2);
        y1 = pk_fma(d2, K2, d1);
        y2 = pk_fma(s2, K4, s1);
        y3 = pk_add(pk_fma(d2, K8, d1), m5);
      };
      f32x2 yp[4][4][2];                           // [row][column][channel pair]
      f32x4 res[4][4];
      auto half = [&](auto h_tag) __attribute__((always_inline)) {
        constexpr int h = decltype(h_tag)::value;
        f32x2 wp[6][4];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          f32x2 m[6];
#pragma unroll
          for (int b = 0; b < 6; ++b) m[b] = f32x2{acc[36 * TH + wf4_slot(a, b)][2 * h], acc[36 * TH + wf4_slot(a, b)][2 * h + 1]};
          at_pk(m[0], m[1], m[2], m[3], m[4], m[5], wp[a][0], wp[a][1], wp[a][2], wp[a][3]);
        }
        if constexpr (HAS_RES) {
          if (h == 0) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((char*)p.res + (size_t)cur.nb * img_px * p.Cout * 4, 0, out_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) res[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, off[i][j], FISR_F4X_SOFF(i, j), 0));
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) at_pk(wp[0][j], wp[1][j], wp[2][j], wp[3][j], wp[4][j], wp[5][j], yp[0][j][h], yp[1][j][h], yp[2][j][h], yp[3][j][h]);
      };
      half(std::integral_constant<int, 0>{});
      half(std::integral_constant<int, 1>{});
      f32x4 pm[2][2];                              // 2x2 max pooling of the lane's 4 x 4 pixels (p.pool_out)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = yp[i][j][0][0], y1 = yp[i][j][0][1], y2 = yp[i][j][1][0], y3 = yp[i][j][1][1];
          f32x4 o;
          o[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y0)));
          o[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y1)));
          o[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y2)));
          o[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_addr, __builtin_bit_cast(int, y3)));
          if constexpr (HAS_RES) o += res[i][j];
          if (p.relu_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, 0, %0" : "+v"(o[e]));
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), os, off[i][j], FISR_F4X_SOFF(i, j), 0);
          if constexpr (POOL) {
            if ((i & 1) == 0 && (j & 1) == 0) pm[i >> 1][j >> 1] = o;
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) pm[i >> 1][j >> 1][e] = fmaxf(pm[i >> 1][j >> 1][e], o[e]);
            }
          }
        }
      if constexpr (POOL) {       // ops.py:54 max_pool 2x2 / 2 of what was just stored: [N, H/2, W/2, Cout]
        const unsigned ph = (unsigned)p.H >> 1, pw = (unsigned)p.W >> 1;
        const unsigned pool_bytes = ph * pw * (unsigned)p.Cout * 4u;
        const __amdgpu_buffer_rsrc_t ps_ = __builtin_amdgcn_make_buffer_rsrc((char*)p.pool_out + (size_t)cur.nb * pool_bytes, 0, pool_bytes, 0x00020000);
        const unsigned pbase = c_ok ? (((unsigned)oy0 >> 1) * pw + ((unsigned)ox0 >> 1)) * (unsigned)p.Cout * 4u + (unsigned)c0 * 4u : OOB;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
          for (int bj = 0; bj < 2; ++bj) {
            const unsigned o_ = (interior || ((oy0 + 2 * bi < p.H) & (ox0 + 2 * bj < p.W))) ? pbase : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pm[bi][bj]), ps_, o_,
                                                   ((unsigned)bi * pw + (unsigned)bj) * (unsigned)p.Cout * 4u, 0);
          }
      }
#undef FISR_F4X_SOFF
    };
    epilogue(std::integral_constant<int, 0>{});
    epilogue(std::integral_constant<int, 1>{});
    if (FISR_F4X_TRACE && p.trace && n_done == 0) t_end1 = __builtin_readcyclecounter();
    if (FISR_F4X_TRACE && p.trace && n_done == 1) t2[5] = __builtin_readcyclecounter();
    ++n_done;
    if (!has_next) break;
    b_cur += gridDim.x;
    cur = nxt;
    has_next = b_cur + (int)gridDim.x < n_items;
    nxt = has_next ? item_of(b_cur + gridDim.x) : cur;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the repeated copies behind the last item still write LDS)
#undef FISR_F4X_MMA1
#undef FISR_F4X_MMA1Z
#undef FISR_F4X_ACC_V
#undef FISR_F4X_DMA1
  if (FISR_F4X_TRACE && p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_main; tr[2] = __builtin_readcyclecounter();
    tr[3] = t_end1; tr[4] = t_first; tr[5] = t_real; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = (unsigned long long)n_done;
  }
  if (FISR_F4X_TRACE && p.trace && (tid & 63) == 0 && n_done > 1) {   // second row block: per WAVE {item start, after iteration 0, 1, K loop end, epilogue start, end}
    unsigned long long* tr = p.trace + ((size_t)gridDim.x + (size_t)blockIdx.x * 8 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) tr[i] = t2[i];
  }
}

#undef FISR_F4X_BEGIN
#undef FISR_F4X_COPY
#undef FISR_F4X_END

// launcher: conv3x3_wf4.h's conditions and grid, 256 threads
inline hipError_t launch_conv_wf4x(const ConvArgs& a, hipStream_t st) {
  static bool attr_done[64] = {};
  static int n_cu[64] = {};
  constexpr size_t lds = wf4_lds_bytes();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    const void* kerns[] = {reinterpret_cast<const void*>(conv3x3_wf4x_kernel<false, false>), reinterpret_cast<const void*>(conv3x3_wf4x_kernel<false, true>),
                           reinterpret_cast<const void*>(conv3x3_wf4x_kernel<true, false>), reinterpret_cast<const void*>(conv3x3_wf4x_kernel<true, true>),
                           reinterpret_cast<const void*>(conv3x3_wf4x_kernel<false, true, true>)};
    for (const void* k : kerns) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipError_t eu = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wf4x_kernel<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)wf4_lds_bytes_ups());
    if (eu != hipSuccess) return eu;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    attr_done[dev] = true;
  }
  const bool plain = a.in0_cs == a.C0 && (a.C1 == 0 || a.in1_cs == a.C1) && a.rec_cs == a.Cout && a.rec_co == 0 && a.slope == 0.f && a.dil == 1;
  if (!plain || !wf4_fits(a.H, a.W, a.C0, a.C1, a.Cout) || (a.res && a.d2s) || a.CoutPad != a.Cout) return hipErrorInvalidValue;
  if (a.pool_out && (a.d2s || (a.H & 1) || (a.W & 1) || a.relu_in || !a.res)) return hipErrorInvalidValue;
  if (a.ups && ((a.H & 1) || (a.W & 1) || a.C1 || a.relu_in || a.res || a.pool_out)) return hipErrorInvalidValue;
  const int items = ((a.W + F4_TW - 1) / F4_TW) * ((a.H + F4_TH - 1) / F4_TH) * a.N * (a.CoutPad / F4_BN);
  const int grid = std::min(items, std::max(8, n_cu[dev] & ~7));
  if (a.ups) {
    hipLaunchKernelGGL((conv3x3_wf4x_kernel<false, false, false, true>), dim3(grid), dim3(256), wf4_lds_bytes_ups(), st, a, items);
  } else if (a.pool_out) {
    hipLaunchKernelGGL((conv3x3_wf4x_kernel<false, true, true>), dim3(grid), dim3(256), lds, st, a, items);
  } else if (a.relu_in) {
    if (a.res) hipLaunchKernelGGL((conv3x3_wf4x_kernel<true, true>), dim3(grid), dim3(256), lds, st, a, items);
    else hipLaunchKernelGGL((conv3x3_wf4x_kernel<true, false>), dim3(grid), dim3(256), lds, st, a, items);
  } else {
    if (a.res) hipLaunchKernelGGL((conv3x3_wf4x_kernel<false, true>), dim3(grid), dim3(256), lds, st, a, items);
    else hipLaunchKernelGGL((conv3x3_wf4x_kernel<false, false>), dim3(grid), dim3(256), lds, st, a, items);
  }
  return hipGetLastError();
}

}  // namespace fisr
