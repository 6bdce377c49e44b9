// 3x3 SAME convolution on the fp16 matrix pipe, fed entirely by LDS-DMA (round 3).
//
// Same operator and fused neighbours as conv3x3.h (reference ops.py:7-11 + relu-on-load / relu / residual / dual-source concat /
// depth_to_space), same NHWC fp16 tensors, same host row order of the weights -- a second KERNEL for the fp16 engine and for the
// fp16 stages of the mixed engine, built with what the fp32 Winograd kernel (conv3x3_wino8p.h) taught:
//
//   * conv3x3.h stages every K chunk global -> registers -> LDS (16-byte registers per thread for the halo tile and for the
//     weight slab, two barriers per chunk) and its wave tile is 2 rows x 64 channels: 10 ds_read_b128 per 12 MFMAs.  At the
//     fp16 rate (one 32x32x16 MFMA = 32 cycles per SIMD, one ds_read_b128 of a wave = 8 cycles of the CU's LDS) that is 83 %
//     of the LDS bandwidth for the fragment reads alone, plus 20 % for the staging writes: the kernel is LDS-bound at 0.31 of
//     the MFMA peak (PMC: MFMA busy 0.42 at 2.13 GHz).
//   * Here a wave owns 4 rows x 32 columns x 64 channels (8 accumulators, 128 registers): per tap column the 6 halo rows are
//     read once and serve all (row, dy) pairs, so a 16-channel chunk costs 18 + 18 fragment reads for 72 MFMAs (0.5 reads per
//     MFMA), and a workgroup's tile is 8 x 64 pixels, so a weight slab is staged once per 512 pixels instead of once per 256.
//   * Every global byte goes global -> LDS by `buffer_load_dwordx4 ... lds` issued from inline asm (hipcc would treat an LDS-DMA it
//     knows about as a FLAT access and drain both counters at every wait): no staging registers, ONE barrier per chunk, the copies
//     of chunk k+1 in flight under the MFMAs of chunk k (two LDS stages).  Buffer loads, because a lane whose offset lies behind
//     the buffer's end gets ZEROS written to LDS (probed on MI355X: scripts/probes/buffer_lds_probe.hip; the scalar offset is
//     part of the range check): the zero padding of SAME convolution costs nothing -- out-of-image halo pixels point there.
//   * LDS records are 32 bytes (16 fp16 channels = one K step of v_mfma_f32_32x32x16_f16) WITHOUT padding, because an LDS-DMA
//     writes 1 KB contiguously per wave instruction; the two 16-byte halves of a record are swapped when bit 3 of the pixel
//     column (weights: of the row) is set, so the 16 lanes of every ds_read_b128 phase hit 16 distinct 16-byte slots.  The DMA
//     applies the swizzle on the SOURCE side (which half a lane fetches); the host bakes it into the weight slabs.
//   * relu-on-load is a v_pk_max_f16 on the fragment registers (4 per fragment, in the MFMAs' shadow): no LDS pass.
//
// Workgroup = 256 threads = 4 waves: wave w -> rows 4*(w&1) .., columns 32*(w>>1) .. of the 8 x 64 tile; two workgroups per CU
// (2 x 79 872 B of LDS, <= 256 registers), which keeps two independent K loops out of phase on every SIMD.
// GENERAL = true adds what PWC-Net's layers need (as in conv3x3_wino8p.h): channel-range input / output of a wider buffer,
// leaky relu, dilation as d x d interleaved sub-images.
#pragma once
#include "conv3x3.h"

namespace fisr {

typedef __attribute__((address_space(3))) void* dma_lds_ptr_t;

constexpr int D_TH = 8, D_TW = 64;                 // pixel tile of a workgroup
constexpr int D_HW = D_TW + 2, D_HH = D_TH + 2;    // halo tile
constexpr int D_REC = 32;                          // bytes per LDS record: 16 fp16 channels
constexpr int D_CH = 16;                           // channels per K chunk
constexpr int D_BN = 64;                           // output channels per workgroup (NT = 2; 32 with NT = 1)
constexpr int D_HALO_UNITS = D_HH * D_HW * 2;      // 16-byte units of a halo chunk: 1320 = 20 full wave copies + 40 lanes
constexpr int D_HALO_COPIES = (D_HALO_UNITS + 63) / 64;            // 21
constexpr int D_HALO_BYTES = D_HALO_COPIES * 1024;                 // 21504: the last copy's idle lanes write zeros into the tail
constexpr int D_W_BYTES = 9 * D_BN * D_REC;                        // 18432: one weight slab (16 channels x 9 taps x 64 rows)
constexpr int D_W_COPIES = D_W_BYTES / 1024;                       // 18
constexpr int D_STAGES = 2;
constexpr size_t dma_lds_bytes() { return (size_t)D_STAGES * (D_HALO_BYTES + D_W_BYTES); }     // 79872 (NT = 2; NT = 1 uses less)

// One LDS-DMA copy: lane l moves 16 bytes from (buffer base + soffset + voffset) to LDS byte address M0 + 16*l; zeros when
// soffset + voffset >= num_records.  Hidden from the compiler (see above); completion is counted by hand (vmcnt).
#define FISR_BLDS_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_BLDS_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen lds\n\t"
#define FISR_BLDS_NEXT(STEP)         "s_add_u32 m0, m0, " #STEP "\n\ts_nop 0\n\t"
#define FISR_BLDS_END(KEEP)          "s_mov_b32 m0, %[" #KEEP "]"

struct dma_stage0_t { static constexpr int value = 0; };
struct dma_stage1_t { static constexpr int value = 1; };

// NT: 32-channel MFMA row tiles per wave = N block of 32 * NT output channels per workgroup.  NT = 1 exists for the flow network's
// layers with Cout = 32 / 96 (a 64-channel block would compute 2x / 1.33x their work); FISRnet's layers all take NT = 2.
template <bool GENERAL, int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_dma_f16_kernel(const ConvArgs p) {
  typedef _Float16 T;
  constexpr int BN = 32 * NT;
  constexpr int WB = 9 * BN * D_REC;                  // bytes of one weight slab: 18432 (NT = 2) / 9216 (NT = 1)
  typedef f16x8 Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [halo stage 0][halo stage 1][weights stage 0][weights stage 1]
  char* const sH = smem;
  char* const sW = smem + D_STAGES * D_HALO_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int rg = wave & 1, cg = wave >> 1;           // row group (rows 4*rg ..), column group (columns 32*cg ..)

  const int dil = GENERAL ? p.dil : 1;
  const int in_cs = GENERAL ? p.in0_cs : p.C0;       // pixel stride of the inputs in elements (both sources: checked by the host)
  const int rec_cs = GENERAL ? p.rec_cs : p.Cout, rec_co = GENERAL ? p.rec_co : 0;
  const int sub_w = (p.W + dil - 1) / dil, sub_h = (p.H + dil - 1) / dil;       // (upper bound of) a sub-image's size
  const int tiles_x = (sub_w + D_TW - 1) / D_TW, tiles_y = (sub_h + D_TH - 1) / D_TH;

  // XCD-aware work order, as in conv3x3.h: workgroup b runs on XCD b % 8; each XCD gets a contiguous range of virtual ids in
  // which the N-blocks of one pixel tile are consecutive (their re-reads of a halo tile hit that XCD's L2)
  int v = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = v & 7, loc = v >> 3;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int nblocks = p.CoutPad / BN;
  int t = v / nblocks;
  const int nblk = v - t * nblocks;
  const int n0 = nblk * BN;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; t /= tiles_y;
  const int sub = GENERAL ? t % (dil * dil) : 0;
  const int nb = GENERAL ? t / (dil * dil) : t;
  const int ry = sub / dil, rx = sub - ry * dil;     // the tile lives in sub-image (ry, rx): pixel (y, x) <-> (y*dil + ry, x*dil + rx)
  const int x0 = tx * D_TW, y0 = ty * D_TH;

  unsigned long long t_start = 0, t_first = 0, t_main = 0, t_real = 0;
  if (p.trace) { t_start = __builtin_readcyclecounter(); t_real = __builtin_amdgcn_s_memrealtime(); }

  // ---- copy geometry: halo copy c (0 .. 20) is issued by wave c & 3 as its copy number c >> 2; lane l of it moves unit 64 c + l
  //      = half (u & 1) of halo pixel u >> 1.  Byte offset of that pixel's record in image nb, or "behind the end" (zeros).
  constexpr unsigned OOB = 0x80000000u;
  unsigned hoff[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int u = 64 * (wave + 4 * q) + lane;
    const int pix = u >> 1;
    const int py = pix / D_HW, px = pix - py * D_HW;
    const int gy = (y0 - 1 + py) * dil + ry, gx = (x0 - 1 + px) * dil + rx;
    const bool ok = u < D_HALO_UNITS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const unsigned half = (unsigned)((u & 1) ^ ((px >> 3) & 1));                 // source-side swizzle
    hoff[q] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)in_cs + half * 8u) * 2u : OOB;
  }
  const size_t img_elems = (size_t)p.H * p.W * in_cs;
  const unsigned img_bytes = (unsigned)(img_elems * 2);
  // (GENERAL: in0 points at the first channel of the range, so the last pixel's record may end past img_bytes - that is
  //  inside the allocation: the range is part of a wider pixel)
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.in0 + (size_t)nb * img_elems), 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in1 ? (const T*)p.in1 + (size_t)nb * img_elems : (const T*)p.in0), 0, img_bytes, 0x00020000);
  const int nch0 = p.C0 / D_CH, nch = (p.C0 + p.C1) / D_CH;
  // weight slabs: [chunk][N block][9 x 64 x 32 B], the LDS image
  const size_t w_bytes = (size_t)nch * nblocks * WB;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (unsigned)w_bytes, 0x00020000);
  const unsigned woff = (unsigned)lane * 16u;
  const unsigned lds_h = (unsigned)(size_t)(dma_lds_ptr_t)sH + (unsigned)wave * 1024u;
  const unsigned lds_w = (unsigned)(size_t)(dma_lds_ptr_t)sW + (unsigned)wave * 1024u;

  auto copy_chunk = [&](int kc, int stage) {
    const bool first = kc < nch0;
    const unsigned so = (unsigned)(first ? kc : kc - nch0) * (unsigned)(D_CH * 2);
    const unsigned lh = lds_h + (unsigned)stage * (unsigned)D_HALO_BYTES;
    unsigned keep;
    // halo: copies wave, wave + 4, ..., < 21 (wave 0: six, the others five)
#define FISR_DMA_HALO(RS)                                                                                                       \
    if (wave == 0) {                                                                                                            \
      asm volatile(FISR_BLDS_BEGIN(keep, lds) FISR_BLDS_COPY(o0, rs, so) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o1, rs, so)      \
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o2, rs, so) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o3, rs, so)          \
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o4, rs, so) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o5, rs, so)          \
                   FISR_BLDS_END(keep)                                                                                          \
                   : [keep] "=&s"(keep) : [rs] "s"(RS), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]),      \
                     [o2] "v"(hoff[2]), [o3] "v"(hoff[3]), [o4] "v"(hoff[4]), [o5] "v"(hoff[5]) : "memory", "scc");             \
    } else {                                                                                                                    \
      asm volatile(FISR_BLDS_BEGIN(keep, lds) FISR_BLDS_COPY(o0, rs, so) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o1, rs, so)      \
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o2, rs, so) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o3, rs, so)          \
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o4, rs, so) FISR_BLDS_END(keep)                                        \
                   : [keep] "=&s"(keep) : [rs] "s"(RS), [so] "s"(so), [lds] "s"(lh), [o0] "v"(hoff[0]), [o1] "v"(hoff[1]),      \
                     [o2] "v"(hoff[2]), [o3] "v"(hoff[3]), [o4] "v"(hoff[4]) : "memory", "scc");                                \
    }
    if (first) { FISR_DMA_HALO(rs0) } else { FISR_DMA_HALO(rs1) }
#undef FISR_DMA_HALO
    // weight slab: 9 * NT linear copies of 1 KB; wave w takes copies w, w + 4, ... (NT = 2: waves 0, 1 five, waves 2, 3 four;
    // NT = 1: wave 0 three, the others two)
    const unsigned sw = (unsigned)(((size_t)kc * nblocks + nblk) * WB);
    const unsigned lw = lds_w + (unsigned)stage * (unsigned)WB;
    const unsigned s0 = sw + (unsigned)wave * 1024u, s1 = s0 + 4096u, s2 = s0 + 8192u, s3 = s0 + 12288u, s4 = s0 + 16384u;
    const int ncopies = (9 * NT - wave + 3) >> 2;
    if (ncopies == 5) {
      asm volatile(FISR_BLDS_BEGIN(keep, lds) FISR_BLDS_COPY(o, rs, s0) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s1)
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s2) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s3)
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s4) FISR_BLDS_END(keep)
                   : [keep] "=&s"(keep) : [rs] "s"(rsw), [lds] "s"(lw), [o] "v"(woff), [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2),
                     [s3] "s"(s3), [s4] "s"(s4) : "memory", "scc");
    } else if (ncopies == 4) {
      asm volatile(FISR_BLDS_BEGIN(keep, lds) FISR_BLDS_COPY(o, rs, s0) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s1)
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s2) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s3)
                   FISR_BLDS_END(keep)
                   : [keep] "=&s"(keep) : [rs] "s"(rsw), [lds] "s"(lw), [o] "v"(woff), [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2),
                     [s3] "s"(s3) : "memory", "scc");
    } else if (ncopies == 3) {
      asm volatile(FISR_BLDS_BEGIN(keep, lds) FISR_BLDS_COPY(o, rs, s0) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s1)
                   FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s2) FISR_BLDS_END(keep)
                   : [keep] "=&s"(keep) : [rs] "s"(rsw), [lds] "s"(lw), [o] "v"(woff), [s0] "s"(s0), [s1] "s"(s1), [s2] "s"(s2) : "memory", "scc");
    } else {
      asm volatile(FISR_BLDS_BEGIN(keep, lds) FISR_BLDS_COPY(o, rs, s0) FISR_BLDS_NEXT(0x1000) FISR_BLDS_COPY(o, rs, s1) FISR_BLDS_END(keep)
                   : [keep] "=&s"(keep) : [rs] "s"(rsw), [lds] "s"(lw), [o] "v"(woff), [s0] "s"(s0), [s1] "s"(s1) : "memory", "scc");
    }
  };
  auto copies_landed_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  copy_chunk(0, 0);                                   // in flight while the accumulators are set up

  // ---- accumulators start from bias (+ residual), as in conv3x3.h: lane (li, kh) of tile [m][j] owns pixel
  //      (y0 + 4 rg + m, x0 + 32 cg + li), channels n0 + 32 j + 16 kh + r
  typedef Rec16<T> R16;
  f32x16 acc[4][NT];
  {
    const bool use_res = p.res != nullptr;
    uint4 rres[4][NT][R16::NV];
    if (use_res) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int y = min((y0 + 4 * rg + m) * dil + ry, p.H - 1), x = min((x0 + 32 * cg + li) * dil + rx, p.W - 1);   // clamped: never stored when outside
        const size_t gp = (size_t)(nb * p.H + y) * p.W + x;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c0 = min(n0 + 32 * j + 16 * kh, p.Cout - 16);
          const uint4* q = reinterpret_cast<const uint4*>((const T*)p.res + gp * rec_cs + rec_co + c0);
#pragma unroll
          for (int k = 0; k < R16::NV; ++k) rres[m][j][k] = q[k];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4* bq = reinterpret_cast<const f32x4*>(p.bias + n0 + 32 * j + 16 * kh);
      float bv[16];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const f32x4 f = bq[k]; bv[4 * k] = f.x; bv[4 * k + 1] = f.y; bv[4 * k + 2] = f.z; bv[4 * k + 3] = f.w; }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        if (use_res) R16::decode(rres[m][j], rv);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][j][r] = bv[r] + rv[r];
      }
    }
  }

  // ---- fragment addresses: pixel column c = 32 cg + li + dx of halo row r sits at (r * 66 + c) * 32 + ((kh ^ bit3(c)) * 16) ----
  const char* a_base[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int c = 32 * cg + li + dx;
    a_base[dx] = sH + ((4 * rg) * D_HW + c) * D_REC + ((kh ^ ((c >> 3) & 1)) * 16);
  }
  const char* const b_base = sW + li * D_REC + ((kh ^ ((li >> 3) & 1)) * 16);
  const uint32_t relu_floor_in = p.relu_in ? 0u : 0xfc00fc00u;

  // One chunk = 3 tap columns x 3 taps x 8 MFMAs.  Fragments are requested one block (8 MFMAs = 256 cycles) ahead of their use:
  // the 6 halo rows of the NEXT column while the first tap of this column runs, the weight pair of the NEXT tap while this tap
  // runs -- a wave issues in order, so a fragment read straight in front of its MFMA leaves the pipe idle for an LDS round trip.
  auto load_rows = [&](Frag (&rows)[6], int S, int dx) {
#pragma unroll
    for (int r = 0; r < 6; ++r) rows[r] = *reinterpret_cast<const Frag*>(a_base[dx] + S * D_HALO_BYTES + r * D_HW * D_REC);
  };
  auto relu_rows = [&](Frag (&rows)[6]) {     // relu-on-load: a maximum with (0, 0) or (-inf, -inf), a uniform operand, no branch
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      uint32_t* q = reinterpret_cast<uint32_t*>(&rows[r]);
#pragma unroll
      for (int k = 0; k < 4; ++k) asm("v_pk_max_f16 %0, %1, %0" : "+v"(q[k]) : "s"(relu_floor_in));
    }
  };
  auto load_b = [&](Frag (&b)[NT], int S, int tap) {
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const Frag*>(b_base + S * WB + (tap * BN + 32 * j) * D_REC);
  };
  auto compute = [&](auto stage_tag) {
    constexpr int S = decltype(stage_tag)::value;
    Frag rows[2][6], bf[2][NT];
    load_rows(rows[0], S, 0);
    load_b(bf[0], S, 0);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      __builtin_amdgcn_sched_barrier(0);            // the relu of this column's rows must not be hoisted to their loads
      relu_rows(rows[dx & 1]);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int blk = dx * 3 + dy;
        if (blk + 1 < 9) load_b(bf[(blk + 1) & 1], S, ((blk + 1) % 3) * 3 + (blk + 1) / 3);      // tap index = dy * 3 + dx
        if (dy == 0 && dx + 1 < 3) load_rows(rows[(dx + 1) & 1], S, dx + 1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[blk & 1][j], rows[dx & 1][m + dy], acc[m][j], 0, 0, 0);
        // pin the interleave.  First tap of a column: the relu of the row an MFMA pair needs (4 VALU) in front of it, the rest
        // and the LDS reads (next column's rows, next tap's weights) in the MFMAs' shadow; other taps: MFMAs and reads only.
        if (dy == 0) {
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, NT == 1 ? 2 : 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, NT == 1 ? 6 : 3, 0);
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4 * NT; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
    }
  };

  // ---- K loop: one barrier per 16-channel chunk; the copies of chunk k+1 fly under the MFMAs of chunk k ----
  copies_landed_barrier();                            // chunk 0 staged
  if (p.trace) t_first = __builtin_readcyclecounter();
  for (int kc = 0; kc < nch; kc += 2) {
    if (kc + 1 < nch) copy_chunk(kc + 1, 1);
    compute(dma_stage0_t{});
    copies_landed_barrier();                          // stage 1 landed, everybody is done reading stage 0
    if (kc + 1 >= nch) break;
    if (kc + 2 < nch) copy_chunk(kc + 2, 0);
    compute(dma_stage1_t{});
    copies_landed_barrier();
  }
  if (p.trace) t_main = __builtin_readcyclecounter();

  // ---- epilogue: relu, fp16 records, 16-byte stores straight from the accumulators (conv3x3.h's record addressing) ----
  const float relu_floor = p.relu_out ? 0.f : -__builtin_huge_valf();
  const float lk = GENERAL ? p.slope : 0.f;
  const bool leaky = GENERAL && p.relu_out && lk != 0.f;
  auto act = [&](float v_) { return fmaxf(v_, leaky ? lk * v_ : relu_floor); };
  const int cq_shift = p.d2s_shift;
  const int x = (x0 + 32 * cg + li) * dil + rx;
  // Stores: a lane's 16-channel record is 32 bytes = two 16-byte units.  Written by the lane alone (two instructions, each a
  // 16-byte piece per pixel) the launch put 1.58 x its output bytes on the HBM write path (r05, WRITE_SIZE: 1.31 GB for 0.83 GB at
  // 64 -> 64 @ 12 x 544 x 992; partial 32-byte sectors from non-temporal 16-byte pieces).  Now neighbouring lanes trade one unit
  // (DPP quad_perm [1,0,3,2]): instruction 0 writes the EVEN lane's record -- the even lane its first unit, the odd lane the
  // second --, instruction 1 the odd lane's, so every instruction writes whole 32-byte records (64 contiguous bytes with the
  // kh = 1 half of the wave).  No lane leaves before the exchange: validity travels with the offsets.
  const bool odd = (li & 1) != 0;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int y = (y0 + 4 * rg + m) * dil + ry;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c0 = n0 + 32 * j + 16 * kh;
      const bool ok = y < p.H && x < p.W && c0 < p.Cout;
      float vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vv[r] = act(acc[m][j][r]);
      uint4 q[R16::NV];
      R16::encode(vv, q);
      size_t e;
      if (p.d2s) {
        const int sb = c0 >> cq_shift, c = c0 & ((1 << cq_shift) - 1);
        e = (((size_t)(nb * 2 * p.H + 2 * y + (sb >> 1))) * (2 * p.W) + 2 * x + (sb & 1)) * ((size_t)1 << cq_shift) + c;
      } else {
        e = ((size_t)(nb * p.H + y) * p.W + x) * rec_cs + rec_co + c0;
      }
      static_assert(R16::NV == 2, "fp16 records are two 16-byte units");
      // the unit this lane gives away: the even lane its second, the odd lane its first
      const uint4 give = odd ? q[0] : q[1];
      uint4 got;
      got.x = dpp_quad_xor1(give.x); got.y = dpp_quad_xor1(give.y); got.z = dpp_quad_xor1(give.z); got.w = dpp_quad_xor1(give.w);
      const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32), okw = ok ? 1u : 0u;
      const uint32_t plo = dpp_quad_xor1(elo), phi = dpp_quad_xor1(ehi), pok = dpp_quad_xor1(okw);
      const size_t e_pair = ((size_t)phi << 32) | plo;
      // instruction 0: the even lane's record (pixel A), instruction 1: the odd lane's (pixel B); this lane's unit = its parity
      const size_t eA = odd ? e_pair : e, eB = odd ? e : e_pair;
      const bool okA = odd ? pok != 0 : ok, okB = odd ? ok : pok != 0;
      const uint4 uA = odd ? got : q[0], uB = odd ? q[1] : got;
      const int uo = odd ? 8 : 0;                         // elements: the second 16-byte unit of a record
#ifndef FISR_DMA_STORE_NT
#define FISR_DMA_STORE_NT 0      // A/B hook: 1 = non-temporal stores (r03's choice)
#endif
      if (okA) {
        u32x4_t nv; nv.x = uA.x; nv.y = uA.y; nv.z = uA.z; nv.w = uA.w;
        if (FISR_DMA_STORE_NT) __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t*>((T*)p.out + eA + uo));
        else *reinterpret_cast<u32x4_t*>((T*)p.out + eA + uo) = nv;
      }
      if (okB) {
        u32x4_t nv; nv.x = uB.x; nv.y = uB.y; nv.z = uB.z; nv.w = uB.w;
        if (FISR_DMA_STORE_NT) __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t*>((T*)p.out + eB + uo));
        else *reinterpret_cast<u32x4_t*>((T*)p.out + eB + uo) = nv;
      }
    }
  }
  if (p.trace && tid == 0) {
    unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
    tr[0] = t_start; tr[1] = t_main; tr[2] = __builtin_readcyclecounter();
    tr[3] = 0; tr[4] = t_first; tr[5] = t_real; tr[6] = __builtin_amdgcn_s_memrealtime(); tr[7] = 0;
  }
}

#undef FISR_BLDS_BEGIN
#undef FISR_BLDS_COPY
#undef FISR_BLDS_NEXT
#undef FISR_BLDS_END

}  // namespace fisr
