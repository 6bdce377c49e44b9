// 3x3 SAME convolution in fp32 by Winograd minimal filtering F(2x2, 3x3) on the gfx950 matrix cores.
//
// Same operator and the same fused neighbours as conv3x3.h (reference ops.py:7-11 + relu / residual / concat /
// depth_to_space), same NHWC fp32 activation tensors -- a second ALGORITHM for the fp32 engine, the one
// cuDNN picks for 3x3 stride-1 fp32 convolutions under the reference's TensorFlow 1.13 (README.md:27-33):
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplies per 2x2 outputs instead of 36: the fp32 MFMA pipe (157 TF/s, the same rate as the vector
// ALU) does 4/9 of the direct algorithm's work; everything is fp32 (transforms are adds/subs in fp32, the
// products accumulate in fp32 on v_mfma_f32_32x32x2_f32); U = G g G^T is computed once on the host in
// double and rounded to fp32.
//
// GEMM view, per transform position p = 0..15:  M_p[co][wtile] = sum_ci U_p[co][ci] * V_p[ci][wtile].
// Work item: output tile 8 rows x 32 cols of pixels = 4 x 16 Winograd tiles ("wtiles") x 64 output channels.  The
// weights are the MFMA row operand and the host packs the rows so that a lane owns 16 consecutive channels (one
// 64-byte record of the activation tensor), exactly as in conv3x3.h.  This header holds what the Winograd kernels share
// (constants, the LDS-DMA macros); the shipped kernel is conv3x3_wino8p.h.  Its two predecessors (one wave per SIMD;
// two waves per SIMD, one item per workgroup) live under diag/ and are compiled into -DFISR_DIAG builds only.
//
// K loop over 8-channel chunks, ONE barrier per chunk, all of LDS in use (163 712 of 163 840 bytes):
//   RAW[3]  (8+2)x(32+2) halo pixels x 32 B          LDS-DMA from the activation tensor, three chunks ahead
//   V[2]    16 positions x 64 wtiles x 32 B          B^T d B of the NEXT chunk, computed from RAW while the MFMAs
//                                                    of the current chunk run
//   U[2]    16 positions x 64 channels x 32 B        LDS-DMA of the host-made slab (its final LDS image)
// Nothing is staged through registers: every global byte goes global -> LDS by global_load_lds_dwordx4.
// The pipeline depths are explicit: a raw chunk has two whole MFMA phases (~8k cycles) to arrive from HBM, a weight
// slab one (L2 hit).
// LDS records are 32 B (8 fp32); the two 16-byte halves of record i are swapped when bit 3 of i is set, so
// the 16 lanes of every ds_read_b128 service group hit 16 distinct 16-byte slots (conflict-free fragment
// reads without padding; MI355X_MICROARCH.md, LDS table).
#pragma once
#include "conv3x3.h"

namespace fisr {

constexpr int W_CH = 8;                        // channels per K chunk
constexpr int W_REC = 32;                      // bytes per LDS record (8 fp32)
constexpr int W_BN = 64;                       // output channels per workgroup
constexpr int W_NWT = 64;                      // Winograd tiles per workgroup (4 x 16 over the 8 x 32 pixel tile)
constexpr int W_SLAB = 16 * 64 * W_REC;        // one V or U buffer: 32768 B
constexpr int W_RAW = HALO_PIX * W_REC;        // one raw halo buffer: 10880 B
constexpr int W_RAW_UNITS = HALO_PIX * 2;      // 16-byte units of a raw halo chunk: 680 = 10 full waves + 40 lanes
constexpr size_t wino_lds_bytes() { return (size_t)4 * W_SLAB + 3 * W_RAW; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// FISR_WABL: performance-diagnosis ablations of this kernel (WRONG results): 1 no input transform in the loop,
// 2 no copies in the loop, 4 no MFMAs in the loop, 8 no fragment reads in the loop (stage 0's are reused),
// 16 no scheduler interleave requests, 32 no padding fix, 64 no output transform (raw accumulators stored).
#ifndef FISR_WABL
#define FISR_WABL 0
#endif

// One LDS-DMA copy instruction per FISR_GLDS_COPY: lane l moves 16 bytes from (gbase + voff) to LDS byte address
// M0 + 16*l.  Written in inline asm ON PURPOSE: hipcc treats a global_load_lds it knows about as a FLAT access
// pending on both counters and then drains vmcnt(0) / lgkmcnt(0) at every later wait and puts a vmcnt wait in
// front of the next ds_read (it cannot disambiguate LDS addresses).  Hidden from the compiler, the copies are
// ordered by the counted s_waitcnt of this file alone -- and there is no compiler-tracked VMEM load in the
// K loop whose wait the hidden copies could falsify.
#define FISR_GLDS_BEGIN(KEEP, LDS)  "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_GLDS_COPY(OFF, G)      "global_load_lds_dwordx4 %[" #OFF "], %[" #G "]\n\t"
// (the raw activation chunks; -DFISR_RAW_AUX_MODE=1|2|3: A/B hook for cache-policy bits on that stream -- nt / sc1 / sc0 sc1)
#ifndef FISR_RAW_AUX_MODE
#define FISR_RAW_AUX_MODE 0
#endif
#if FISR_RAW_AUX_MODE == 1
#define FISR_RAW_AUX " nt"
#elif FISR_RAW_AUX_MODE == 2
#define FISR_RAW_AUX " sc1"
#elif FISR_RAW_AUX_MODE == 3
#define FISR_RAW_AUX " sc0 sc1"
#else
#define FISR_RAW_AUX ""
#endif
#define FISR_GLDS_COPY_RAW(OFF, G)  "global_load_lds_dwordx4 %[" #OFF "], %[" #G "]" FISR_RAW_AUX "\n\t"
#define FISR_GLDS_NEXT_ROW          "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
#define FISR_GLDS_END(KEEP)         "s_mov_b32 m0, %[" #KEEP "]"

}  // namespace fisr
