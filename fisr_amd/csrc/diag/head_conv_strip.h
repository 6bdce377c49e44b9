// The 3- and 6-channel heads of the fp32 engine (FISRnet.py:100,105; head_conv.h) as a STRIP WALK fed by LDS-DMA (round 5).
//
// What the counters say about head_conv.h (profiles/pmc_traffic.json, r04): its launches move 4.2 GB (3-channel head) and 6.2 GB
// (6-channel head) of HBM for 3.1 GB of algorithmic bytes, at 5.2 and 6.2 TB/s -- the kernel IS at the memory system's rate, on
// 1.35 x / 2.0 x the bytes: an 8 x 32 tile re-reads a third of its input as halo (10 x 34 pixels for 8 x 32 outputs) and nothing of
// that hits L2 (the vertical neighbour runs 5 MB of input later), and the 6-channel head's 16-channel chunks fetch every 128-byte
// line twice, a chunk apart.  So here the input is read ONCE, in whole 256-byte pixel records:
//   * a workgroup owns a strip of 30 output columns (32 with the halo = eight groups of four pixels) and walks down it in steps of
//     8 rows.  LDS holds a ring of 18 rows x 32 pixels x 64 channels: a step reads 10 rows (8 new + the 2 carried over from the
//     step before), the 8 rows of the NEXT step land meanwhile in the 8 slots the step before last has left.  Re-read: 32 / 30
//     horizontally, two rows per 136-row segment vertically: 1.08 x the tensor instead of 1.33 x (2.0 x).
//   * every byte goes global -> LDS by `buffer_load_dwordx4 ... lds` (inline asm, hand-counted waits: conv3x3_dma.h), one 1-KB
//     copy = four pixels x 64 channels, so the memory pipeline sees whole lines; out-of-image pixels point behind the buffer's end
//     and arrive as zeros (the SAME padding).  A copy writes 1 KB contiguously, so records cannot be padded; instead a group's 1 KB
//     holds [channel quad 16][pixel 4][16 bytes] and the groups of a row are 1088 bytes apart: the 16 lanes of a ds_read_b128 phase
//     (16 consecutive pixels, one quad) sit at 16 x (pixel mod 16) bytes mod 256 -- sixteen different bank groups for any
//     alignment of the tap window -- and the quad index is an immediate offset (no address arithmetic in the loop).
//   * 512 threads: lanes 0-255 sum channels 0-31 of "their" pixel, lanes 256-511 channels 32-63 (weights stay uniform per wave:
//     scalar operands, as in head_conv.h), two waves per SIMD so that one wave's LDS / scalar-load round trips hide under the
//     other's FMAs (one workgroup per CU: the ring is 153 KB); the halves meet through 6 KB of LDS at the end of a step.
// Arithmetic: v_pk_fma_f32 acc[o, o+1] += x * w[o, o+1] as in head_conv.h, summed per channel half (taps outer, channels inner),
// the halves added last -- a different summation order than head_conv.h's chunk-major one, equal to rounding (parity tests:
// tests/test_gpu_parity.py, against the fp64 oracle).
#pragma once
#include "../head_conv.h"

namespace fisr {

constexpr int HS_OW = 30, HS_HW = 32, HS_R = 8, HS_RING = 18;      // output columns of a strip, with halo; rows per step; ring rows
constexpr int HS_G = 1088, HS_ROW = (HS_HW / 4) * HS_G;            // bytes per 4-pixel group (1 KB + 64) and per ring row (8704)
constexpr int HS_RING_BYTES = HS_RING * HS_ROW;                    // 156672
constexpr int HS_PART_BYTES = 256 * 3 * 8;                         // partial sums of the upper channel half: 256 pixels x <= 3 pairs
constexpr size_t head_strip_lds_bytes() { return (size_t)HS_RING_BYTES + HS_PART_BYTES; }      // 162816 of 163840
constexpr int HS_CIN = 64;

#define FISR_HS_BEGIN(KEEP, LDS)   "s_mov_b32 %[" #KEEP "], m0\n\ts_mov_b32 m0, %[" #LDS "]\n\ts_nop 0\n\t"
#define FISR_HS_COPY(OFF, RS, SO)  "buffer_load_dwordx4 %[" #OFF "], %[" #RS "], %[" #SO "] offen lds\n\t"
#define FISR_HS_END(KEEP)          "s_mov_b32 m0, %[" #KEEP "]"

template <int NPAIR, bool RELU_IN>
__global__ __launch_bounds__(512) void head_conv_strip_kernel(const HeadArgs p, const int seg_rows, const int n_items) {
  extern __shared__ __attribute__((aligned(16))) char hs[];
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  typedef const float __attribute__((address_space(4))) * cptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;                               // channel half of this wave's lanes
  const int pix = tid & 255, px = pix & 31, py = pix >> 5;  // this lane's pixel of the 8 x 32 step (columns 30, 31: no output)
  const int strips_x = (p.W + HS_OW - 1) / HS_OW, segs_y = (p.H + seg_rows - 1) / seg_rows;
  const size_t img_bytes = (size_t)p.H * p.W * HS_CIN * 4;
  const unsigned ring0 = (unsigned)(size_t)(lds_ptr_t)hs;
  f2* const part = reinterpret_cast<f2*>(hs + HS_RING_BYTES);
  constexpr unsigned OOB = 0x80000000u;
  // column part of the tap addresses: pixel column hc = min(px, 29) + dx of the halo'd strip -> (hc / 4) groups + (hc % 4) slots
  int co[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hc = (px < HS_OW ? px : HS_OW - 1) + dx;
    co[dx] = (hc >> 2) * HS_G + (hc & 3) * 16 + half * 512;      // (+ the channel half: quads 8 half .. 8 half + 7, 64 bytes each)
  }
  const cptr_t wbase = (cptr_t)(unsigned long long)(p.w + (size_t)half * 32 * (2 * NPAIR));
  const cptr_t bias = (cptr_t)(unsigned long long)p.bias;

  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    // item -> (image, strip, segment), segments of a strip first: vertical neighbours run back to back
    int t = item;
    const int sy = t % segs_y; t /= segs_y;
    const int sx = t % strips_x;
    const int nb = t / strips_x;
    const int x0 = sx * HS_OW, ys = sy * seg_rows, ye = min(ys + seg_rows, p.H);
    const int nsteps = (ye - ys + HS_R - 1) / HS_R;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + (size_t)nb * img_bytes), 0, (unsigned)img_bytes, 0x00020000);
    // this wave copies group `wave` of every row: lane -> pixel (lane & 3) of the group, channel quad lane >> 2
    unsigned coff;
    {
      int l = lane;
      asm volatile("" : "+v"(l));
      const int gx = x0 - 1 + 4 * wave + (l & 3);
      coff = (gx >= 0 && gx < p.W) ? (unsigned)gx * (unsigned)(HS_CIN * 4) + (unsigned)(l >> 2) * 16u : OOB;
    }
    // ring rows k = 0 .. : image row ys - 1 + k lives in slot k % 18
    auto copy_rows = [&](int k0, int nrows, int slot0) __attribute__((always_inline)) {
      int slot = slot0;
      for (int i = 0; i < nrows; ++i) {
        const int gy = ys - 1 + k0 + i;
        const bool row_ok = gy >= 0 && gy < p.H;
        const unsigned so = row_ok ? (unsigned)gy * (unsigned)p.W * (unsigned)(HS_CIN * 4) : 0u;
        const unsigned vo = row_ok ? coff : OOB;
        const unsigned lds = ring0 + (unsigned)slot * (unsigned)HS_ROW + (unsigned)wave * (unsigned)HS_G;
        unsigned keep;
        asm volatile(FISR_HS_BEGIN(keep, lds) FISR_HS_COPY(o, rs, so) FISR_HS_END(keep)
                     : [keep] "=&s"(keep) : [rs] "s"(rs), [so] "s"(so), [lds] "s"(lds), [o] "v"(vo) : "memory", "scc");
        slot = slot + 1 == HS_RING ? 0 : slot + 1;
      }
    };
    copy_rows(0, HS_R + 2, 0);
    int slot_s = 0;                                         // slot of ring row 8 s
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's copies of the step's rows have landed (and its stores of the step before are through)
      __builtin_amdgcn_s_barrier();                                     // ... everyone's; and everyone has left the step before
      asm volatile("" ::: "memory");
      if (s + 1 < nsteps) {
        int sl = slot_s + HS_R + 2;
        sl = sl >= HS_RING ? sl - HS_RING : sl;
        copy_rows(HS_R * s + HS_R + 2, HS_R, sl);            // the next step's 8 new rows, under this step's FMAs
      }
      f2 acc[NPAIR];
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) acc[k] = half == 0 ? f2{bias[2 * k], bias[2 * k + 1]} : f2{0.f, 0.f};
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) {
        int sl = slot_s + py + dy;
        sl = sl >= HS_RING ? sl - HS_RING : sl;
        const char* row = hs + sl * HS_ROW;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const char* rec = row + co[dx];
          // uniform per wave -> scalar loads; through the CONSTANT address space (behind the copies' asm "memory" clobber a plain
          // global load is no longer provably unclobbered and becomes one vector load per weight: head_conv_dma.h)
          const cptr_t wt = wbase + (size_t)(dy * 3 + dx) * HS_CIN * (2 * NPAIR);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            f32x4 x = *reinterpret_cast<const f32x4*>(rec + q * 64);
            if (RELU_IN) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
            const f2 xlo = {x.x, x.y}, xhi = {x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const cptr_t we = wt + (4 * q + e) * (2 * NPAIR);
#pragma unroll
              for (int k = 0; k < NPAIR; ++k) {
                const f2 wp = {we[2 * k], we[2 * k + 1]};
                // low half: x_e * w[2k], high half: x_e * w[2k + 1] -- op_sel picks x_e out of its register pair for both
                if ((e & 1) == 0)
                  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[k]) : "v"(e < 2 ? xlo : xhi), "s"(wp));
                else
                  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[k]) : "v"(e < 2 ? xlo : xhi), "s"(wp));
              }
            }
          }
        }
      }
      // the channel halves meet: the upper half parks its sums, the lower half adds them and stores
      if (half == 1) {
#pragma unroll
        for (int k = 0; k < NPAIR; ++k) part[pix * NPAIR + k] = acc[k];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (half == 0) {
        const int x = x0 + px, y = ys + HS_R * s + py;
        if (px < HS_OW && x < p.W && y < ye) {
          float* ob = p.out + ((size_t)(nb * p.H + y) * p.W + x) * (size_t)p.out_cstride;
#pragma unroll
          for (int k = 0; k < NPAIR; ++k) {
            const f2 o = part[pix * NPAIR + k];
            acc[k].x += o.x; acc[k].y += o.y;
          }
#pragma unroll
          for (int n = 0; n < 2 * NPAIR; ++n)
            if (n < p.Cout) {
              float v = (n & 1) ? acc[n >> 1].y : acc[n >> 1].x;
              if (p.relu_out) v = fmaxf(v, 0.f);
              ob[n + p.out_coff + (n >= p.out_split ? p.out_gap : 0)] = v;
            }
        }
      }
      slot_s += HS_R;
      slot_s = slot_s >= HS_RING ? slot_s - HS_RING : slot_s;
    }
    // (the next item's first copies overwrite slots 0-9: every wave has left the last step's FMAs -- the barrier above)
  }
}

#undef FISR_HS_BEGIN
#undef FISR_HS_COPY
#undef FISR_HS_END

// what the strip kernel takes: 64 input channels (256-byte pixel records), 31-bit byte offsets inside an image
inline bool head_strip_fits(int h, int w, int cin) { return cin == HS_CIN && (double)h * w * HS_CIN * 4.0 < 2147483648.0; }

inline hipError_t launch_head_strip(const HeadArgs& h, hipStream_t st) {
  static bool attr_done[64] = {};
  static int n_cu[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    const void* ks[] = {reinterpret_cast<const void*>(head_conv_strip_kernel<2, false>), reinterpret_cast<const void*>(head_conv_strip_kernel<2, true>),
                        reinterpret_cast<const void*>(head_conv_strip_kernel<3, false>), reinterpret_cast<const void*>(head_conv_strip_kernel<3, true>)};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)head_strip_lds_bytes());
      if (e != hipSuccess) return e;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    attr_done[dev] = true;
  }
  if (!head_strip_fits(h.H, h.W, h.Cin)) return hipErrorInvalidValue;
  // segments: as long as possible (each start re-reads two rows and pays an exposed round trip) while every CU still gets >= 4 items
  const int strips = (h.W + HS_OW - 1) / HS_OW;
  int seg = 136;
  while (seg > 24 && (long long)strips * ((h.H + seg - 1) / seg) * h.N < 4LL * n_cu[dev]) seg = (seg / 2 + 7) & ~7;
  const int items = strips * ((h.H + seg - 1) / seg) * h.N;
  const int grid = std::min(items, n_cu[dev]);
  const size_t lds = head_strip_lds_bytes();
  if (h.Cout <= 4) {
    if (h.relu_in) hipLaunchKernelGGL((head_conv_strip_kernel<2, true>), dim3(grid), dim3(512), lds, st, h, seg, items);
    else hipLaunchKernelGGL((head_conv_strip_kernel<2, false>), dim3(grid), dim3(512), lds, st, h, seg, items);
  } else {
    if (h.relu_in) hipLaunchKernelGGL((head_conv_strip_kernel<3, true>), dim3(grid), dim3(512), lds, st, h, seg, items);
    else hipLaunchKernelGGL((head_conv_strip_kernel<3, false>), dim3(grid), dim3(512), lds, st, h, seg, items);
  }
  return hipGetLastError();
}

}  // namespace fisr
