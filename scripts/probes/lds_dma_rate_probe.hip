// Probe (diagnostics, not product): what one CU's vector-memory path delivers, bytes per clock, for
//   mode 0  buffer_load_dwordx4 ... lds   (LDS-DMA), a wave copy = 1 KB contiguous
//   mode 1  buffer_load_dwordx4 -> VGPRs, the same addresses
//   mode 2  LDS-DMA, 64-byte pieces at a 256-byte stride (four lanes per pixel record: the halo copies of conv3x3_dma_fs.h)
//   mode 3  VGPR loads of those pieces
// from an L2-resident footprint (every workgroup walks the same 256 KB) and from a streaming one (each workgroup its own range of
// a 2 GB buffer).  8 waves per CU (two workgroups of 256), 10 copies in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate(const char* src, size_t wg_stride, unsigned span, int iters, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * wg_stride), 0, span, 0x00020000);
  const bool strided = MODE >= 2;
  // contiguous: lane l of copy c reads 16 B at (c * 64 + l) * 16; strided: 16 B at ((c * 16 + (l >> 2)) * 256 + (l & 3) * 16)
  unsigned voff = strided ? (unsigned)((lane >> 2) * 256 + (lane & 3) * 16) : (unsigned)lane * 16u;
  const unsigned step = strided ? 16u * 256u : 1024u;         // per copy
  const unsigned lbase = (unsigned)(size_t)(lds_ptr_t)lds + (unsigned)wave * 10240u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 accv = {0, 0, 0, 0};
  unsigned so = (unsigned)wave * 10u * step;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      unsigned keep, s = so;
      asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_add_u32 %[s], %[s], %[st]\n\t"
                   "buffer_load_dwordx4 %[o], %[rs], %[s] offen lds\n\t"
                   "s_mov_b32 m0, %[keep]\n\ts_waitcnt vmcnt(0)"
                   : [keep] "=&s"(keep), [s] "+s"(s) : [rs] "s"(rs), [lds] "s"(lbase), [o] "v"(voff), [st] "s"(step) : "memory", "scc");
    } else {
      u32x4 v[10];
#pragma unroll
      for (int c = 0; c < 10; ++c) v[c] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + c * step, 0);
#pragma unroll
      for (int c = 0; c < 10; ++c) accv ^= v[c];
    }
    so += 40u * step;                                   // the workgroup's four waves cover 40 copies per round
    if (so + 10u * step > span) so = (unsigned)wave * 10u * step;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (accv.x == 0x12345u && accv.y == 7u) out[threadIdx.x] = 1.f;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const size_t big = (size_t)2 << 30;
  char* d; float* o; unsigned long long* c;
  hipMalloc(&d, big); hipMalloc(&o, 4096); hipMalloc(&c, 8 * 4096);
  hipMemset(d, 1, big);
  const int blocks = 512, iters = 400;
  const char* names[] = {"LDS-DMA, 1 KB contiguous per copy", "VGPR loads, 1 KB contiguous", "LDS-DMA, 64-byte pieces / 256-byte stride", "VGPR loads, 64-byte pieces / 256-byte stride"};
  for (int foot = 0; foot < 2; ++foot) {
    // L2-resident: every workgroup the same 256 KB; streaming: 4 MB per workgroup, walked once (400 rounds x 40 copies x 1 KB = 16 MB contiguous: wraps 4 x;
    // the strided form spans 4 x the bytes it reads)
    const size_t wg_stride = foot == 0 ? 0 : (size_t)4 << 20;
    const unsigned span = foot == 0 ? 256u << 10 : 4u << 20;
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 40960, 0, d, wg_stride, span, iters, o, c);
        if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 40960, 0, d, wg_stride, span, iters, o, c);
        if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 40960, 0, d, wg_stride, span, iters, o, c);
        if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 40960, 0, d, wg_stride, span, iters, o, c);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> hc(blocks);
      hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : hc) s += (double)v; s /= blocks;
      const double bytes_wg = (double)iters * 40 * 1024;
      printf("%-10s %-46s %7.1f B/clk per workgroup, %7.1f per CU (two workgroups)   [%0.f cycles per round of 10 copies per wave]\n",
             foot == 0 ? "L2-hot" : "streaming", names[mode], bytes_wg / s, 2 * bytes_wg / s, s / iters);
    }
  }
  return 0;
}
