// 3x3 SAME convolution in the fp16 + fp8 split format ("f16f8", conv3x3.h) fed entirely by LDS-DMA, persistent (round 5).
//
// Same operator and fused neighbours as conv3x3.h (reference ops.py:7-11 + relu-on-load / relu / residual / dual-source concat /
// depth_to_space / the 2x2 max pooling of ops.py:54 as a second store), same NHWC f16f8 tensors.  What round 4 measured about the
// register-staged kernel in this format: a 64->64 full-resolution workgroup lives 46 k cycles of which 11 k are its
// prologue (the first HBM round trips) and 9 k its epilogue, the fragment reads of its 2-row wave tile need 105 % of the LDS
// bandwidth at the MFMA rate, and it sits at 248 of 256 registers with its staging registers -- 0.34 of the matrix pipe.  A
// 16-channel chunk of this format is 64-byte records: 42 KB of halo and 37 KB of weights, so a CU's LDS holds TWO chunks.
//
//   * Two workgroups per CU, ONE LDS stage each: the double buffering is done by the hardware's interleaving of the two
//     workgroups -- while one waits for its next chunk the other multiplies -- and the same interleaving hides each one's
//     epilogue (the f16f8 encoder is ~240 vector instructions per record) and every stall of a wave's own stream.  (Measured
//     first, r05: one workgroup per CU with two stages and one wave per SIMD -- the K loop took 9-10 k cycles per chunk for 4864
//     MFMA cycles and the 12-20 k cycles of an epilogue were paid in full: no faster than the old kernel.)
//   * Every global byte goes global -> LDS by `buffer_load_dwordx4 ... lds` (inline asm, zero padding by range check:
//     conv3x3_dma.h): no staging registers, which is what makes room for the wave tile below at two waves per SIMD.
//   * persistent: grid = 2 x CUs; a workgroup walks the items blockIdx.x, + gridDim.x, ... of an XCD-aware order
//     (conv3x3_wino8p.h); the first chunk of the next item is requested before the epilogue of this one and lands under it.
//   * wave tile 4 rows x 32 columns x 64 channels (8 accumulators), 4 waves = 8 x 64 pixels: per tap column the 6 halo rows are
//     read once and serve all (row, dy) pairs -- for the fp8 cross terms too, which is what the r05 record format is for: a
//     lane's fp8 operand of one tap is ONE 16-byte unit, {l8 | h8} of 8 channels against {wh8 | wl8} of the same 8 channels (both
//     cross terms of a channel in one K block under one scale pair), so a fragment belongs to a pixel, not to a tap pair, and a
//     block-scaled MFMA takes its two K blocks from two fragments: taps (dy 0, dy 1) of a column = halo rows m, m + 1 (adjacent
//     registers), (dy 2, dx 0 | dx 1), and (dy 2, dx 2) beside a zero block.  72 fragment reads per 72 fp16 + 40 fp8 MFMAs of a
//     chunk (4864 MFMA cycles per wave): 47 % of the LDS bandwidth at the MFMA rate (the old kernel: 80 reads per 2432 cycles).
//     One register set per fragment kind: the h rows of the next column are requested while the fp8 MFMAs of this one run, the
//     fp8 rows while the fp16 MFMAs run.
//   * LDS halo records are the tensor's 64-byte records, copied by FOUR neighbouring lanes (one 64-byte request per pixel: the
//     memory pipeline charges for requests); the four 16-byte units of a record are XOR-rotated by bits 2-3 of the
//     pixel column on the SOURCE side of the copy, so the 16 lanes of every ds_read_b128 phase (16 consecutive columns, one unit)
//     hit 16 distinct slots.  Weight slabs are the host-made LDS image: [9 taps][64 rows][32 B of w_h] (conv3x3_dma.h's) + five
//     fp8 groups [plane = 2 kh + tap of the pair][64 rows][16 B].
//   * relu-on-load on the fragment registers: v_pk_max_f16 for h, a byte mask from h8's own sign bits for {l8 | h8} (12 VALU).
//   * residual: loaded in the epilogue four lanes per record (the stores' quad layout), one record ahead, transposed by DPP.
#pragma once
#include "conv3x3_dma.h"

namespace fisr {

constexpr int FS_TH = 8, FS_HH = FS_TH + 2;             // rows of a workgroup's pixel tile; its width TW is 64 or 32 (below)
constexpr int FS_BN = 64;                                // output channels per item
constexpr int FS_TRACE_WORDS = 16;                // 8-byte words of a workgroup's trace record (diagnostics: p.trace)
constexpr int FS_CH = 16;                                // channels per K chunk = one 64-byte record
constexpr int FS_REC = 64;
constexpr int fs_halo_units(int tw) { return FS_HH * (tw + 2) * 4; }               // 16-byte units of a halo chunk: 2640 / 1360
constexpr int fs_halo_copies(int tw) { return (fs_halo_units(tw) + 63) / 64; }     // 42 / 22 wave copies of 1 KB
constexpr int fs_halo_bytes(int tw) { return fs_halo_copies(tw) * 1024; }          // 43008 / 22528
constexpr int FS_WM_BYTES = 9 * FS_BN * 32;                            // 18432: w_h, conv3x3_dma.h's image
constexpr int FS_WX_BYTES = 4 * 4096 + 2048;                           // 18432: four tap pairs x 4 planes + tap 8 x 2 planes
constexpr int FS_W_BYTES = FS_WM_BYTES + FS_WX_BYTES;                  // 36864
constexpr int FS_W_COPIES = FS_W_BYTES / 1024;                         // 36: nine per wave
// FISR_FS_HALO2 (r06, narrow tile only): TWO halo stages beside the one weight slab -- 2 x 22 528 + 36 864 = 81 920 B, exactly half of a
// CU's LDS, so two workgroups per CU still fit.  The halo of chunk k + 1 (HBM) is requested BEFORE chunk k is multiplied and flies
// under its MFMAs; only the weight slab (L2) is requested behind the barrier that releases it.  Same-box A/B (two alternating rounds):
// 64 -> 64 @ 12 x 544 x 992 1449 -> 1407 / 1492 -> 1451 us, 48 -> 64 1121 -> 1084, 64 -> 64 @ 272 x 496 370 -> 355; `mixed` step 61.4 -> 60.8 ms
// (113.9 -> 115.2 frames/s).  (One copy per MFMA block INSIDE the multiply phase was tried behind it and faulted; not pursued.)
// -DFISR_FS_HALO2=0: the one-stage form, for A/B runs.
#ifndef FISR_FS_HALO2
#define FISR_FS_HALO2 1
#endif
constexpr bool fs_halo2(int tw) { return FISR_FS_HALO2 != 0 && tw == 32; }
constexpr size_t dmafs_lds_bytes(int tw) { return (size_t)fs_halo_bytes(tw) * (fs_halo2(tw) ? 2 : 1) + FS_W_BYTES; }      // 79872 / 59392 (81920): two workgroups per CU

#define FISR_FS_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_FS_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen lds\n\t"
#define FISR_FS_NEXT               "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
#define FISR_FS_NEXTW(SO)          "s_add_u32 m0, m0, 0x1000\n\ts_add_u32 %[" #SO "], %[" #SO "], 0x1000\n\t"
#define FISR_FS_END(KEEP)          "s_mov_b32 m0, %[" #KEEP "]"

// The record encoder of Rec16<fsplit> without its 48 clamps per record: with MODE.FP16_OVFL set (the kernel sets it at its start)
// v_cvt_pk_f16_f32 saturates to +-65504 and the fp8 conversions to +-448 by themselves (probed on MI355X:
// scripts/probes/fp16_ovfl_cvt_probe.hip) -- the same bytes as fsplit_encode8 for every finite input.
// r06: 40 vector instructions per record instead of 72 (the epilogue is a fifth to a third of an item: 8-10.6 k cycles of vector work).
//   * the remainder v - h comes from ONE v_fma_mix_f32 per value (h read as the f16 half it is: no f16 -> f32 conversion, exact);
//   * its scaling by 2^14 is the scale operand of v_cvt_scalef32_pk_fp8_f32 (which DIVIDES by it: 2^-14), no multiply;
//   * the fp8 copy of h is v_cvt_scalef32_pk_fp8_f16 straight from the packed halves.
// Bit-identical to the form above over 4 M random records of magnitudes 2^-22 .. 2^17, zeros, +-1e30 and values beyond fp16's range
// (scripts/probes/f8_encoder_probe.hip).  -DFISR_FS_ENC=0: the r05 form, for A/B runs.
#ifndef FISR_FS_ENC
#define FISR_FS_ENC 1
#endif
__device__ __forceinline__ void fsplit_encode16_sat(const float* v, uint4* q) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
#if FISR_FS_ENC
  typedef short s2_t __attribute__((ext_vector_type(2)));
  uint32_t hw[8];
  float r[16];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    f2_t f; f.x = v[2 * d]; f.y = v[2 * d + 1];
    hw[d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, h2_t));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[2 * d]) : "v"(hw[d]), "v"(v[2 * d]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[2 * d + 1]) : "v"(hw[d]), "v"(v[2 * d + 1]));
  }
  q[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  q[1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
  const float sr = 1.f / (float)(1 << FS_LSHIFT);
#pragma unroll
  for (int g = 0; g < 2; ++g) {      // channels 8 g .. 8 g + 7: {l8 | h8}
    s2_t t;
    uint4 x;
    t = s2_t{0, 0};
    t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, r[8 * g], r[8 * g + 1], sr, false); t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, r[8 * g + 2], r[8 * g + 3], sr, true);
    x.x = __builtin_bit_cast(uint32_t, t);
    t = s2_t{0, 0};
    t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, r[8 * g + 4], r[8 * g + 5], sr, false); t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, r[8 * g + 6], r[8 * g + 7], sr, true);
    x.y = __builtin_bit_cast(uint32_t, t);
    t = s2_t{0, 0};
    t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(t, __builtin_bit_cast(h2_t, hw[4 * g]), 1.f, false); t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(t, __builtin_bit_cast(h2_t, hw[4 * g + 1]), 1.f, true);
    x.z = __builtin_bit_cast(uint32_t, t);
    t = s2_t{0, 0};
    t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(t, __builtin_bit_cast(h2_t, hw[4 * g + 2]), 1.f, false); t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(t, __builtin_bit_cast(h2_t, hw[4 * g + 3]), 1.f, true);
    x.w = __builtin_bit_cast(uint32_t, t);
    q[2 + g] = x;
  }
#else
  uint32_t hw[8];
  float hf[16], r[16];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    f2_t f; f.x = v[2 * d]; f.y = v[2 * d + 1];
    const h2_t h = __builtin_convertvector(f, h2_t);
    hw[d] = __builtin_bit_cast(uint32_t, h);
    hf[2 * d] = (float)h.x; hf[2 * d + 1] = (float)h.y;
    r[2 * d] = (v[2 * d] - hf[2 * d]) * (float)(1 << FS_LSHIFT);
    r[2 * d + 1] = (v[2 * d + 1] - hf[2 * d + 1]) * (float)(1 << FS_LSHIFT);
  }
  q[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  q[1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
#pragma unroll
  for (int g = 0; g < 2; ++g) {      // channels 8 g .. 8 g + 7: {l8 | h8}
    int t;
    uint4 x;
    t = __builtin_amdgcn_cvt_pk_fp8_f32(r[8 * g], r[8 * g + 1], 0, false); t = __builtin_amdgcn_cvt_pk_fp8_f32(r[8 * g + 2], r[8 * g + 3], t, true); x.x = (uint32_t)t;
    t = __builtin_amdgcn_cvt_pk_fp8_f32(r[8 * g + 4], r[8 * g + 5], 0, false); t = __builtin_amdgcn_cvt_pk_fp8_f32(r[8 * g + 6], r[8 * g + 7], t, true); x.y = (uint32_t)t;
    t = __builtin_amdgcn_cvt_pk_fp8_f32(hf[8 * g], hf[8 * g + 1], 0, false); t = __builtin_amdgcn_cvt_pk_fp8_f32(hf[8 * g + 2], hf[8 * g + 3], t, true); x.z = (uint32_t)t;
    t = __builtin_amdgcn_cvt_pk_fp8_f32(hf[8 * g + 4], hf[8 * g + 5], 0, false); t = __builtin_amdgcn_cvt_pk_fp8_f32(hf[8 * g + 6], hf[8 * g + 7], t, true); x.w = (uint32_t)t;
    q[2 + g] = x;
  }
#endif
}

// TW: width of the workgroup's pixel tile.  64: wave (rg, cg) owns rows 4 rg .., columns 32 cg .., all 64 channels (8 accumulators:
//     every pixel fragment feeds two MFMAs).  32: wave (rg, nh) owns rows 4 rg .., all 32 columns, channels 32 nh .. (4 accumulators).
//     The narrow tile is for the layers with ONE 64-channel output block: a 16-channel chunk is HALF of a 128-byte line of the
//     NHWC tensor, the other half follows a chunk later, and what must survive in L2 meanwhile is every resident workgroup's
//     halo -- 64 workgroups per XCD x 84 KB of lines with the wide tile: 5.4 MB against 4 MB of L2, measured as 2.8 x the input's
//     bytes fetched from HBM (r05, FETCH_SIZE); 2.8 MB with the narrow one.  Layers with several output blocks share their lines
//     between the blocks of a tile (neighbours in the XCD's range) and keep the wide tile.
// RELU_IN: conv(relu(x)); HAS_RES: + residual (plain layout, may alias out); POOL: the 2x2 maxima of the output as a second store
template <int TW, bool RELU_IN, bool HAS_RES, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_dma_fs_kernel(const ConvArgs p, const int n_items) {
  typedef fsplit T;
  typedef Rec16<T> R16;
  constexpr int FS_TW = TW, FS_HW = TW + 2;
  constexpr int FS_HALO_UNITS = fs_halo_units(TW), FS_HALO_COPIES = fs_halo_copies(TW), FS_HALO_BYTES = fs_halo_bytes(TW);
  constexpr int NQ = (FS_HALO_COPIES + 3) / 4;         // halo copies of a wave
  constexpr int NJ = TW / 32;                          // 32-channel output blocks of a wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [halo chunk][weight slab]
  constexpr bool H2 = fs_halo2(TW);
  char* const sH = smem;
  char* const sW = smem + FS_HALO_BYTES * (H2 ? 2 : 1);
  int hst = 0;                                         // H2: byte offset of the halo stage the chunk being multiplied lives in

  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");      // MODE.FP16_OVFL: the f16 / fp8 conversions saturate
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int rg = wave & 1;                           // row group (rows 4*rg ..)
  const int cg = TW == 64 ? wave >> 1 : 0;           // column group (columns 32*cg ..)
  const int j0 = TW == 64 ? 0 : wave >> 1;           // first 32-channel block of this wave

  const int tiles_x = (p.W + FS_TW - 1) / FS_TW, tiles_y = (p.H + FS_TH - 1) / FS_TH;
  const int nblocks = p.CoutPad / FS_BN;
  const int nch0 = p.C0 / FS_CH, nch = (p.C0 + p.C1) / FS_CH;

  // work item b (0 .. n_items-1) -> (x0, y0, nb, nblk): XCD-aware order as in conv3x3_wino8p.h -- workgroup w runs on XCD
  // w % 8, gridDim.x is a multiple of 8 (or n_items), so the items of a workgroup stay on one XCD, each XCD walks a contiguous
  // range of tiles and the N blocks of a tile are neighbours in it
  struct Item { int x0, y0, nb, nblk; };
  auto item_of = [&](int b) {
    const int q = n_items >> 3, r = n_items & 7;
    const int xcd = b & 7, loc = b >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    int t = v / nblocks;
    Item it;
    it.nblk = v - t * nblocks;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    it.nb = t / tiles_y;
    it.x0 = tx * FS_TW; it.y0 = ty * FS_TH;
    return it;
  };

  unsigned long long t_start = 0, t_real = 0, c_k = 0, c_ep = 0, c_wait = 0, c_comp = 0, c_bar = 0, c_issue = 0;
  if (p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  // ---- copy geometry: halo copy c (0 .. FS_HALO_COPIES - 1) is issued by wave c & 3 as its copy number c >> 2; lane l of it fills LDS
  //      slot u = 64 c + l = slot (u & 3) of halo pixel u >> 2, with unit (u & 3) ^ bits 2-3 of the pixel column.
  constexpr unsigned OOB = 0x80000000u;
  const unsigned pix_bytes = (unsigned)p.C0 * 4u;      // pixel stride of the inputs in bytes (both sources: checked by the host)
  const size_t img_in = (size_t)p.H * p.W * pix_bytes;
  const size_t w_bytes = (size_t)nch * nblocks * FS_W_BYTES;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (unsigned)w_bytes, 0x00020000);
  const unsigned woff = (unsigned)lane * 16u;
  const unsigned lds_h = (unsigned)(size_t)(dma_lds_ptr_t)sH + (unsigned)wave * 1024u;
  const unsigned lds_w = (unsigned)(size_t)(dma_lds_ptr_t)sW + (unsigned)wave * 1024u;

  // The halo offsets of an item: kept in registers across its chunks by the narrow tile (six registers, ~150 vector instructions
  // saved per chunk); the wide tile, which has no register to spare, recomputes them per chunk -- ALL of them before the first
  // copy: a compiler-made memory access between two copies (a spill reload) waits for vmcnt(0), i.e. for every copy in flight
  // (measured at 13.7 k cycles per chunk).
  constexpr bool KEEP_GEOM = TW == 32;
  unsigned hoff_item[NQ];
  auto halo_geom = [&](unsigned (&hoff)[NQ], const Item& it) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int u = 64 * (wave + 4 * q) + lane;
      const int pix = u >> 2;
      const int py = pix / FS_HW, px = pix - py * FS_HW;
      const int gy = it.y0 - 1 + py, gx = it.x0 - 1 + px;
      const bool ok = u < FS_HALO_UNITS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const unsigned unit = (unsigned)((u & 3) ^ ((px >> 2) & 3));
      hoff[q] = ok ? (unsigned)(gy * p.W + gx) * pix_bytes + unit * 16u : OOB;
    }
  };
  // chunk kc of item `it` -> LDS (KEEP_GEOM: hoff_item describes `it`)
  auto copy_chunk = [&](int kc, const Item& it, bool do_halo = true, bool do_w = true, int stage_off = 0) {
    const int nb = it.nb, nblk = it.nblk;
    const bool first = kc < nch0;
    const unsigned so = (unsigned)(first ? kc : kc - nch0) * (unsigned)FS_REC;
    const char* src = first ? (const char*)p.in0 : (const char*)p.in1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)nb * img_in), 0, (unsigned)img_in, 0x00020000);
    unsigned keep;
    unsigned hoff[NQ];
    if constexpr (KEEP_GEOM) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) hoff[q] = hoff_item[q];
    } else {
      halo_geom(hoff, it);
    }
    const unsigned lh = lds_h + (unsigned)stage_off;
    if (do_halo) {
    // the wave's copies: NQ of them, or NQ - 1 for the waves behind the last partial round
    constexpr int LASTW = FS_HALO_COPIES - 4 * (NQ - 1);      // waves < LASTW issue NQ copies
#define FISR_FS_H1(Q)  FISR_FS_COPY(o##Q, rs, so)
#define FISR_FS_HN(Q)  FISR_FS_NEXT FISR_FS_COPY(o##Q, rs, so)
    if constexpr (NQ == 11) {
      if (wave < LASTW)
        asm volatile(FISR_FS_BEGIN(keep, lds) FISR_FS_H1(0) FISR_FS_HN(1) FISR_FS_HN(2) FISR_FS_HN(3) FISR_FS_HN(4) FISR_FS_HN(5) FISR_FS_HN(6) FISR_FS_HN(7)
                     FISR_FS_HN(8) FISR_FS_HN(9) FISR_FS_HN(10) FISR_FS_END(keep)
                     : [keep] "=&s"(keep)
                     : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]), [o2] "v"(hoff[2]), [o3] "v"(hoff[3]),
                       [o4] "v"(hoff[4]), [o5] "v"(hoff[5]), [o6] "v"(hoff[6]), [o7] "v"(hoff[7]), [o8] "v"(hoff[8]), [o9] "v"(hoff[9]),
                       [o10] "v"(hoff[NQ - 1])
                     : "memory", "scc");
      else
        asm volatile(FISR_FS_BEGIN(keep, lds) FISR_FS_H1(0) FISR_FS_HN(1) FISR_FS_HN(2) FISR_FS_HN(3) FISR_FS_HN(4) FISR_FS_HN(5) FISR_FS_HN(6) FISR_FS_HN(7)
                     FISR_FS_HN(8) FISR_FS_HN(9) FISR_FS_END(keep)
                     : [keep] "=&s"(keep)
                     : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]), [o2] "v"(hoff[2]), [o3] "v"(hoff[3]),
                       [o4] "v"(hoff[4]), [o5] "v"(hoff[5]), [o6] "v"(hoff[6]), [o7] "v"(hoff[7]), [o8] "v"(hoff[8]), [o9] "v"(hoff[9])
                     : "memory", "scc");
    } else {
      static_assert(NQ == 6 || NQ == 11, "halo copy count");
      if (wave < LASTW)
        asm volatile(FISR_FS_BEGIN(keep, lds) FISR_FS_H1(0) FISR_FS_HN(1) FISR_FS_HN(2) FISR_FS_HN(3) FISR_FS_HN(4) FISR_FS_HN(5) FISR_FS_END(keep)
                     : [keep] "=&s"(keep)
                     : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]), [o2] "v"(hoff[2]), [o3] "v"(hoff[3]),
                       [o4] "v"(hoff[4]), [o5] "v"(hoff[5])
                     : "memory", "scc");
      else
        asm volatile(FISR_FS_BEGIN(keep, lds) FISR_FS_H1(0) FISR_FS_HN(1) FISR_FS_HN(2) FISR_FS_HN(3) FISR_FS_HN(4) FISR_FS_END(keep)
                     : [keep] "=&s"(keep)
                     : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]), [o2] "v"(hoff[2]), [o3] "v"(hoff[3]),
                       [o4] "v"(hoff[4])
                     : "memory", "scc");
    }
#undef FISR_FS_H1
#undef FISR_FS_HN
    }
    if (!do_w) return;
    // weight slab: 36 linear copies of 1 KB, wave w takes copies w, w + 4, ... (nine each)
    unsigned sw = (unsigned)(((size_t)kc * nblocks + nblk) * FS_W_BYTES) + (unsigned)wave * 1024u;
    const unsigned lw = lds_w;
    asm volatile(FISR_FS_BEGIN(keep, lds) FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s) FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s)
                 FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s) FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s) FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s)
                 FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s) FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s) FISR_FS_COPY(o, rs, s) FISR_FS_NEXTW(s)
                 FISR_FS_COPY(o, rs, s) FISR_FS_END(keep)
                 : [keep] "=&s"(keep), [s] "+s"(sw)
                 : [rs] "s"(rsw), [lds] "s"(lw), [o] "v"(woff)
                 : "memory", "scc");
  };
  auto copies_landed_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- fragment addresses: pixel column c = 32 cg + li + dx of halo row r is
  //      record r * 66 + c; the main-term unit of this lane is kh (channels 8 kh ..), its fp8 unit 2 + kh, both rotated by bits 2-3 of c
  const char* a_main[3];
  const char* a_x[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = 32 * cg + li + dx;
    const int rot = (c >> 2) & 3;
    const char* rec = sH + ((4 * rg) * FS_HW + c) * FS_REC;
    a_main[dx] = rec + ((kh ^ rot) * 16);
    a_x[dx] = rec + (((2 + kh) ^ rot) * 16);
  }
  const char* const b_main = sW + (32 * j0 + li) * 32 + ((kh ^ ((li >> 3) & 1)) * 16);
  const char* const b_x = sW + FS_WM_BYTES + kh * 2048 + (32 * j0 + li) * 16;        // pair q: + 4096 q, tap t of the pair: + 1024 t, j: + 512 j
  const char* const b_x8 = sW + FS_WM_BYTES + 4 * 4096 + kh * 1024 + (32 * j0 + li) * 16;
  const int sa = 127 - FS_LSHIFT, sb = 127 - p.wexp;

  typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
  auto as_h = [](const uint4& v) { return __builtin_bit_cast(h8_t, v); };
  auto cat = [](const uint4& a, const uint4& b) {
    i32x8 r;
    r[0] = (int)a.x; r[1] = (int)a.y; r[2] = (int)a.z; r[3] = (int)a.w; r[4] = (int)b.x; r[5] = (int)b.y; r[6] = (int)b.z; r[7] = (int)b.w;
    return r;
  };

  f32x16 acc[4][NJ];

  // One chunk = 72 fp16 + 40 block-scaled fp8 MFMAs of this wave, in 14 blocks of 8 (one MFMA per accumulator).  A wave issues in
  // order and a 16-bit MFMA only shadows what is issued BEHIND it while it runs (32 / 64 cycles), so every block carries the
  // fragment reads and the vector work of LATER blocks between its MFMAs (sched_group_barrier pins one MFMA, then its share of
  // reads / vector instructions).  One register set per fragment kind:
  //   column dx:  M0 (dy 0)  reads: weights of dy 1, the column's fp8 rows X        (X is free since the column before)
  //               M1 (dy 1)  reads: weights of dy 2, the fp8 weights of (dy 0, dy 1); vector: relu of X rows 0-2
  //               M2 (dy 2)  reads: the fp8 weights of the dy 2 group;               vector: relu of X rows 3-5
  //               F  (fp8 dy 0 | dy 1: K block 0 = halo row m, block 1 = row m + 1)  reads: the next column's h rows H and its first
  //                                                                                  weights (H is free); vector: relu of H, late
  //               G  (fp8 dy 2: (dx 0 | dx 1) in column 1, dx 2 beside a zero weight block in column 2)
  auto compute = [&]() {
    auto ld_h = [&](uint4 (&H)[6], int dx) {
#pragma unroll
      for (int r = 0; r < 6; ++r) H[r] = *reinterpret_cast<const uint4*>(a_main[dx] + (H2 ? hst : 0) + r * (FS_HW * FS_REC));
    };
    auto ld_x = [&](uint4 (&X)[6], int dx) {
#pragma unroll
      for (int r = 0; r < 6; ++r) X[r] = *reinterpret_cast<const uint4*>(a_x[dx] + (H2 ? hst : 0) + r * (FS_HW * FS_REC));
    };
    auto ld_bm = [&](uint4 (&B)[NJ], int tap) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) B[j] = *reinterpret_cast<const uint4*>(b_main + (tap * FS_BN + 32 * j) * 32);
    };
    auto ld_wx = [&](i32x8 (&Wp)[NJ], int q) {      // both taps of pair q
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        Wp[j] = cat(*reinterpret_cast<const uint4*>(b_x + q * 4096 + j * 512), *reinterpret_cast<const uint4*>(b_x + q * 4096 + 1024 + j * 512));
    };
    auto relu_h = [&](uint4 (&H)[6]) {
      if constexpr (RELU_IN) {
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 6; ++r) H[r] = __builtin_bit_cast(uint4, __builtin_elementwise_max(__builtin_bit_cast(f16x8, H[r]), z));
      }
    };
    auto relu_x = [&](uint4 (&X)[6], int r0) {
      if constexpr (RELU_IN) {
#pragma unroll
        for (int r = r0; r < r0 + 3; ++r) fsplit_relu_x(X[r]);
      }
    };
    // the order inside a block: one MFMA, then `nds` LDS reads and `nva` vector instructions, eight times
#define FISR_FS_SCHED(NDS, NVA)                                                   \
    _Pragma("unroll") for (int g_ = 0; g_ < 4 * NJ; ++g_) {                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
      if ((NDS) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (NDS), 0);       \
      if ((NVA) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NVA), 0);       \
    }                                                                             \
    __builtin_amdgcn_sched_barrier(0)
    uint4 H[6], X[6], Y0[4], Bm[2][NJ];
    i32x8 Wa[NJ], Wb[NJ];
    ld_h(H, 0);
    ld_bm(Bm[0], 0);
    relu_h(H);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      // ---- M0: dy = 0 (tap dx)
      ld_bm(Bm[1], 3 + dx);
      ld_x(X, dx);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(Bm[0][j]), as_h(H[m]), acc[m][j], 0, 0, 0);
      FISR_FS_SCHED(NJ == 2 ? 1 : 2, 0);
      // ---- M1: dy = 1
      ld_bm(Bm[0], 6 + dx);
      ld_wx(Wa, dx);
      relu_x(X, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(Bm[1][j]), as_h(H[m + 1]), acc[m][j], 0, 0, 0);
      FISR_FS_SCHED(NJ == 2 ? 1 : 2, RELU_IN ? (NJ == 2 ? 5 : 10) : 0);
      // ---- M2: dy = 2
      if (dx == 1) ld_wx(Wb, 3);
      if (dx == 2) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < NJ; ++j) Wb[j] = cat(z, *reinterpret_cast<const uint4*>(b_x8 + j * 512));
      }
      relu_x(X, 3);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(Bm[0][j]), as_h(H[m + 2]), acc[m][j], 0, 0, 0);
      FISR_FS_SCHED(NJ == 2 ? 1 : 2, RELU_IN ? (NJ == 2 ? 5 : 10) : 0);
      // ---- F: cross terms of taps (dy 0, dy 1) of this column; the h rows are free: the next column's are requested (first half
      //      of the block) and relu'd (second half), and its first weights
      if (dx < 2) { ld_h(H, dx + 1); ld_bm(Bm[0], dx + 1); }
      if (dx == 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) Y0[m] = X[m + 2];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Wa[j], cat(X[m], X[m + 1]), acc[m][j], 0, 0, 0, sb, 0, sa);
      if (dx < 2) relu_h(H);
      if (dx < 2) {
#pragma unroll
        for (int g_ = 0; g_ < 2 * NJ; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NJ == 2 ? 2 : 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
#pragma unroll
        for (int g_ = 0; g_ < 2 * NJ; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, RELU_IN ? (NJ == 2 ? 10 : 18) : 4, 0);
        }
      } else {
#pragma unroll
        for (int g_ = 0; g_ < 4 * NJ; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- G: cross terms of the dy = 2 taps
      if (dx == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Wb[j], cat(Y0[m], X[m + 2]), acc[m][j], 0, 0, 0, sb, 0, sa);
      } else if (dx == 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Wb[j], cat(X[m + 1], X[m + 2]), acc[m][j], 0, 0, 0, sb, 0, sa);
      }
      if (dx >= 1) {
#pragma unroll
        for (int g_ = 0; g_ < 4 * NJ; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef FISR_FS_SCHED
  };

  // ---- epilogue of one item: (+ residual) -> relu -> f16f8 records -> quad-transposed 64-byte stores (conv3x3.h's record store)
  const float relu_floor = p.relu_out ? 0.f : -__builtin_huge_valf();
  const unsigned img_out = (unsigned)((size_t)p.H * p.W * p.Cout * 4);          // bytes of one output image (plain and d2s layout alike)
  const int cq_shift = p.d2s_shift;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  auto epilogue = [&](const Item& it) {
    const int n0 = it.nblk * FS_BN;
    const int xq = it.x0 + 32 * cg + (li & ~3);        // first pixel of this lane's quad
    const int x = it.x0 + 32 * cg + li;
    const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (size_t)it.nb * img_out, 0, img_out, 0x00020000);
    // residual: record (j, m) is requested while record (j, m) - 1 is converted and stored (four lanes per 64-byte record)
    uint4 rq[2][4];
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(HAS_RES ? (char*)p.res + (size_t)it.nb * img_out : (char*)p.out, 0, img_out, 0x00020000);
    auto ld_res = [&](uint4 (&q)[4], int j, int m) {
      const int y = it.y0 + 4 * rg + m;
      const unsigned c0 = (unsigned)(n0 + 32 * (j0 + j) + 16 * kh);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned eo = (((unsigned)(y * p.W) + (unsigned)(xq + k)) * (unsigned)p.Cout + c0) * 4u + 16u * (unsigned)(li & 3);
        const unsigned off = (y < p.H && xq + k < p.W) ? eo : img_out;
#ifndef FISR_FS_RES_AUX
#define FISR_FS_RES_AUX 2        // the residual is read once: a non-temporal load keeps it from ageing the halo lines out of L2 (r05: FETCH_SIZE
#endif                           // of 64 -> 64 + residual at 12 x 544 x 992 4.69 -> 4.26 GB, mixed step 62.49 -> 62.11 ms; A/B hook: 0)
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, FISR_FS_RES_AUX);
        q[k] = make_uint4(v.x, v.y, v.z, v.w);
      }
    };
    if constexpr (HAS_RES) ld_res(rq[0], 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c0 = n0 + 32 * (j0 + j) + 16 * kh;
      const unsigned sub = (unsigned)c0 >> cq_shift;
      const unsigned eB = p.d2s ? 2u << cq_shift : (unsigned)p.Cout;
#pragma unroll
      for (int mp = 0; mp < 2; ++mp) {
        float pv[16];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int m = 2 * mp + mm;
          const int y = it.y0 + 4 * rg + m;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[m][j][r];
          if constexpr (HAS_RES) {
            constexpr int dummy_ = 0; (void)dummy_;
            const int idx = j * 4 + m;
            if (idx + 1 < 4 * NJ) ld_res(rq[(idx + 1) & 1], (idx + 1) >> 2, (idx + 1) & 3);
            quad_transpose(rq[idx & 1], lane);
            float rv[16];
            R16::decode(rq[idx & 1], rv);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += rv[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], relu_floor);
          if constexpr (POOL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = mm == 0 ? v[r] : fmaxf(pv[r], v[r]);
          }
          uint4 q[4];
          fsplit_encode16_sat(v, q);
          quad_transpose(q, lane);
          const unsigned e0 = p.d2s ? ((((unsigned)(2 * y) + (sub >> 1)) * (unsigned)(2 * p.W) + (sub & 1u)) << cq_shift) + ((unsigned)c0 & ((1u << cq_shift) - 1u))
                                    : (unsigned)(y * p.W) * (unsigned)p.Cout + (unsigned)c0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned eo = (e0 + (unsigned)(xq + k) * eB) * 4u + 16u * (unsigned)(li & 3);
            const unsigned off = (y < p.H && xq + k < p.W) ? eo : img_out;
            u32x4_t nv; nv.x = q[k].x; nv.y = q[k].y; nv.z = q[k].z; nv.w = q[k].w;
            __builtin_amdgcn_raw_buffer_store_b128(nv, os, off, 0, 2);      // streaming store
          }
        }
        if constexpr (POOL) {
          // ops.py:54: a pooling window = two rows of this wave x a lane pair; the even lane stores the pooled record
          const int y = it.y0 + 4 * rg + 2 * mp;
#pragma unroll
          for (int r = 0; r < 16; ++r) pv[r] = fmaxf(pv[r], __uint_as_float(dpp_quad_xor1(__float_as_uint(pv[r]))));
          if (!(li & 1) && y < p.H && x < p.W) {
            uint4 q[4];
            fsplit_encode16_sat(pv, q);
            uint4* ob = reinterpret_cast<uint4*>((char*)p.pool_out + (((size_t)(it.nb * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1)) * p.Cout + c0) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) ob[k] = q[k];
          }
        }
      }
    }
  };

  // ---- the workgroup's items.  One LDS stage: a chunk is requested when everybody has read the one before it, and the
  //      workgroup waits for it -- the CU's other workgroup multiplies meanwhile; the next item's first chunk lands under the epilogue.
  int b_cur = blockIdx.x;
  if (b_cur >= n_items) return;
  Item cur = item_of(b_cur);
  if constexpr (KEEP_GEOM) halo_geom(hoff_item, cur);
  copy_chunk(0, cur);
  int n_done = 0;
  for (;;) {
    const int b_nxt = b_cur + (int)gridDim.x;
    const bool has_next = b_nxt < n_items;
    const Item nxt = has_next ? item_of(b_nxt) : cur;
    {   // accumulators start from the bias: lane (li, kh) of tile [m][j] owns pixel (y0 + 4 rg + m, x0 + 32 cg + li), channels n0 + 32 j + 16 kh + r
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4* bq = reinterpret_cast<const f32x4*>(p.bias + cur.nblk * FS_BN + 32 * (j0 + j) + 16 * kh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 f = bq[k];
#pragma unroll
          for (int m = 0; m < 4; ++m) { acc[m][j][4 * k] = f.x; acc[m][j][4 * k + 1] = f.y; acc[m][j][4 * k + 2] = f.z; acc[m][j][4 * k + 3] = f.w; }
        }
      }
    }
    unsigned long long t0 = 0, t1 = 0;
    if (p.trace) t0 = __builtin_readcyclecounter();
    for (int kc = 0; kc < nch; ++kc) {
      unsigned long long t2 = 0, t3 = 0, t4 = 0;
      if (p.trace) t1 = __builtin_readcyclecounter();
      copies_landed_barrier();            // this chunk has landed
      if (p.trace) { t2 = __builtin_readcyclecounter(); c_wait += t2 - t1; }
      if constexpr (H2) {
        // the next chunk's halo -> the other stage (last read a chunk ago: the barrier above lies behind those reads), before the MFMAs
        if (kc + 1 < nch) copy_chunk(kc + 1, cur, true, false, hst ^ FS_HALO_BYTES);
        else if (has_next) {
          halo_geom(hoff_item, nxt);
          copy_chunk(0, nxt, true, false, hst ^ FS_HALO_BYTES);
        }
      }
      // the wave that multiplies goes first on its SIMD: its partner (the CU's other workgroup) is then in its epilogue or its copy
      // phase -- vector and memory instructions that wait anyway.  `mixed` 111.2 -> 113.3 frames/s (two alternating rounds on one
      // box, priority 1 / 2 / 3 alike; the same on the fp16 kernel's K loop: nothing)
#ifndef FISR_FS_PRIO
#define FISR_FS_PRIO 1
#endif
      __builtin_amdgcn_s_setprio(FISR_FS_PRIO);
      compute();
      __builtin_amdgcn_s_setprio(0);
      if (p.trace) { t3 = __builtin_readcyclecounter(); c_comp += t3 - t2; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();       // everybody is done reading it
      asm volatile("" ::: "memory");
      if (p.trace) { t4 = __builtin_readcyclecounter(); c_bar += t4 - t3; }
      if constexpr (H2) {
        if (kc + 1 < nch) copy_chunk(kc + 1, cur, false, true);
        else if (has_next) copy_chunk(0, nxt, false, true);
        hst ^= FS_HALO_BYTES;
      } else {
        if (kc + 1 < nch) copy_chunk(kc + 1, cur);
        else if (has_next) {
          if constexpr (KEEP_GEOM) halo_geom(hoff_item, nxt);
          copy_chunk(0, nxt);
        }
      }
      if (p.trace) c_issue += __builtin_readcyclecounter() - t4;
    }
    if (p.trace) { t1 = __builtin_readcyclecounter(); c_k += t1 - t0; }
    epilogue(cur);
    if (p.trace) c_ep += __builtin_readcyclecounter() - t1;
    ++n_done;
    if (!has_next) break;
    cur = nxt; b_cur = b_nxt;
  }
  if (p.trace && tid == 0) {      // (FS_TRACE_WORDS words per workgroup: fisr_api.hip sizes the buffer with it, scripts/trace_fs.py reads it)
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * FS_TRACE_WORDS;
    tr[0] = t_start; tr[1] = c_k; tr[2] = __builtin_readcyclecounter();
    tr[3] = c_ep; tr[4] = (unsigned long long)n_done; tr[5] = t_real; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = c_wait;
    tr[8] = c_comp; tr[9] = c_bar; tr[10] = c_issue;
  }
}

#undef FISR_FS_BEGIN
#undef FISR_FS_COPY
#undef FISR_FS_NEXT
#undef FISR_FS_NEXTW
#undef FISR_FS_END

}  // namespace fisr
