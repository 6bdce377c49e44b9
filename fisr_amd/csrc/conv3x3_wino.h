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
// Workgroup = 256 threads = 4 waves (one per SIMD, the kernel uses the whole 512-entry register file):
// output tile 8 rows x 32 cols of pixels = 4 x 16 Winograd tiles ("wtiles") x 64 output channels.  A wave
// owns 32 wtiles x 32 channels for ALL 16 positions = 16 accumulators of the 32x32 MFMA = 256 registers,
// so the output transform A^T M A is done in registers, per lane, with no exchange.  The weights are the
// MFMA row operand and the host packs the rows so that a lane owns 16 consecutive channels (one 64-byte
// record of the activation tensor), exactly as in conv3x3.h.
//
// K loop over 8-channel chunks, ONE barrier per chunk, everything double-buffered in LDS (150 KB):
//   RAW[2]  (8+2)x(32+2) halo pixels x 32 B          global -> registers -> LDS (relu-on-load here)
//   V[2]    16 positions x 64 wtiles x 32 B          B^T d B of the NEXT chunk, computed by all 256 threads
//                                                    from RAW while the MFMAs of the current chunk run
//   U[2]    16 positions x 64 channels x 32 B        straight global -> LDS copies (the host stores the
//                                                    slab in its final LDS image, swizzle included)
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
constexpr int W_RAW_UNITS = HALO_PIX * 2;      // 16-byte units of a raw halo chunk
constexpr size_t wino_lds_bytes() { return (size_t)4 * W_SLAB + 2 * W_RAW; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// GLDS: the U slab goes global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write);
// otherwise it is staged through registers like the raw halo.
template <bool GLDS>
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

  // ---- raw halo loader: unit u = tid + 256*i -> halo pixel u >> 1, 16-byte half u & 1 (= tid & 1) ----
  const int r_half = tid & 1;
  int in_pix[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = tid + 256 * i;
    const int pix = u >> 1;
    const int py = pix / HALO_W, px = pix - py * HALO_W;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = u < W_RAW_UNITS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    in_pix[i] = ok ? (nb * p.H + gy) * p.W + gx : -1;
  }
  auto load_raw = [&](int kc, uint4 (&r)[3]) {
    const float* src;
    int csrc, coff;
    const int c0 = kc * W_CH;
    if (c0 < p.C0) { src = (const float*)p.in0; csrc = p.C0; coff = c0; }
    else           { src = (const float*)p.in1; csrc = p.C1; coff = c0 - p.C0; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      uint4 q = make_uint4(0u, 0u, 0u, 0u);
      if (in_pix[i] >= 0) q = *reinterpret_cast<const uint4*>(src + (size_t)in_pix[i] * csrc + coff + r_half * 4);
      r[i] = q;
    }
  };
  const float relu_in_floor = p.relu_in ? 0.f : -__builtin_huge_valf();   // branch-free relu-on-load
  auto store_raw = [&](int buf, const uint4 (&r)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int u = tid + 256 * i;
      if (i < 2 || u < W_RAW_UNITS) {
        f32x4 f = __builtin_bit_cast(f32x4, r[i]);
        f.x = fmaxf(f.x, relu_in_floor); f.y = fmaxf(f.y, relu_in_floor);
        f.z = fmaxf(f.z, relu_in_floor); f.w = fmaxf(f.w, relu_in_floor);
        *reinterpret_cast<f32x4*>(sR + buf * W_RAW + u * 16) = f;
      }
    }
  };
  // ---- U slab: 2048 16-byte units, a linear copy of the host-made LDS image of (chunk kc, N-block) ----
  const char* const u_base = (const char*)p.wpk + (size_t)nblk * W_SLAB;
  const size_t u_stride = (size_t)nblocks * W_SLAB;
  auto load_u = [&](int kc, int buf, uint4 (&r)[8]) {
    const char* g = u_base + (size_t)kc * u_stride;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (GLDS) {
        // LDS destination = wave-uniform base + lane * 16: units (wave*64 + 256*i) .. +63
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g + (size_t)(tid + 256 * i) * 16),
                                         (lds_ptr_t)(sU + buf * W_SLAB + (wave * 64 + 256 * i) * 16), 16, 0, 0);
      } else {
        r[i] = *reinterpret_cast<const uint4*>(g + (size_t)(tid + 256 * i) * 16);
      }
    }
  };
  auto store_u = [&](int buf, const uint4 (&r)[8]) {
    if constexpr (!GLDS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(sU + buf * W_SLAB + (tid + 256 * i) * 16) = r[i];
    }
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
  // The transform is cut in five slices so that it can be spread between the MFMAs of the running chunk:
  //   slice 0: raw reads of columns 0,1      slice 1: raw reads of columns 2,3 + row stage of columns 0,1
  //   slice 2: row stage of columns 2,3      slice 3: column stage + stores of T-row A
  //   slice 4: column stage + stores of T-row B
  f32x4 txa[2], tza[2], tyb[2], tzb[2], TA[4], TB[4];
  auto tr_read = [&](int rbuf, int cpair) {
    const char* rb = sR + rbuf * W_RAW + t_roff;
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
      TA[2 * cpair + k] = txa[k] - tza[k];
      TB[2 * cpair + k] = tyb[k] + sgn * tzb[k];
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
  auto transform = [&](int rbuf, int vbuf) {     // the whole thing at once (prologue)
    tr_read(rbuf, 0); tr_rows(0); tr_read(rbuf, 1); tr_rows(1); tr_cols(vbuf, 0); tr_cols(vbuf, 1);
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

  // ---- prologue: chunk 0 staged and transformed, chunk 1 raw in LDS ----
  uint4 rraw[3], ru[8];
  load_raw(0, rraw);
  load_u(0, 0, ru);
  store_raw(0, rraw);
  store_u(0, ru);
  if (nch > 1) load_raw(1, rraw);
  __syncthreads();
  transform(0, 0);
  if (nch > 1) store_raw(1, rraw);
  // LDS-DMA copies are tracked by vmcnt only: drain them explicitly before the barrier that publishes the slab
  if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (p.trace) t_first = __builtin_readcyclecounter();

  // ---- main loop: MFMAs of chunk kc || input transform of chunk kc+1 || loads of U(kc+1), raw(kc+2) ----
  for (int kc = 0; kc + 1 < nch; ++kc) {
    const int b = kc & 1;
    load_u(kc + 1, b ^ 1, ru);
    const bool more = kc + 2 < nch;
    if (more) load_raw(kc + 2, rraw);
    frag_load(b, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      if (pp < 7) frag_load(b, pp + 1, (pp + 1) & 1);
      // transform of chunk kc+1, RAW[(kc+1)&1] -> V[(kc+1)&1], one slice per stage
      if (pp == 0) tr_read(b ^ 1, 0);
      if (pp == 1) { tr_rows(0); tr_read(b ^ 1, 1); }
      if (pp == 2) tr_rows(1);
      if (pp == 3) tr_cols(b ^ 1, 0);
      if (pp == 4) tr_cols(b ^ 1, 1);
      FISR_WINO_MMA(pp, pp & 1, x) FISR_WINO_MMA(pp, pp & 1, y) FISR_WINO_MMA(pp, pp & 1, z) FISR_WINO_MMA(pp, pp & 1, w)
      FISR_WINO_INTERLEAVE(3)
      __builtin_amdgcn_sched_barrier(0);
    }
    store_u(b ^ 1, ru);
    if (more) store_raw(b, rraw);       // RAW[kc&1] held chunk kc (consumed one iteration ago)
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // epilogue geometry: lane (li, kh) owns wtile 32*wh + li -> pixels (y0 + 2*ty + i, x0 + 2*tx + j), and the
  // 16-channel record c0 .. c0+15.  The residual records (4 pixels x 64 B) and the bias are requested BEFORE the
  // last chunk's MFMAs so their latency hides under them (the residual may alias the output: every element is
  // read and written by the same lane only, so hoisting the reads above the stores is safe).
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
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // s[c][r] = column c of row i of (A^T M):  i = 0: m0c + m1c + m2c,  i = 1: m1c - m2c - m3c
      float s[4][16];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[c][r] = i == 0 ? (acc[0 + c][r] + acc[4 + c][r]) + acc[8 + c][r]
                           : (acc[4 + c][r] - acc[8 + c][r]) - acc[12 + c][r];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float o[16], rv[16];
        Rec16<float>::decode(rres[i][j], rv);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float yv = j == 0 ? (s[0][r] + s[1][r]) + s[2][r] : (s[1][r] - s[2][r]) - s[3][r];
          o[r] = fmaxf((yv + bv[r]) + rv[r], relu_floor);
        }
        if (px_ok[i][j]) {
          uint4 q[4];
          Rec16<float>::encode(o, q);
          typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
          u32x4_t* dst = reinterpret_cast<u32x4_t*>((float*)p.out + record(y0 + 2 * ty + i, x0 + 2 * tx + j));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            u32x4_t nv; nv.x = q[k].x; nv.y = q[k].y; nv.z = q[k].z; nv.w = q[k].w;
            __builtin_nontemporal_store(nv, dst + k);
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
