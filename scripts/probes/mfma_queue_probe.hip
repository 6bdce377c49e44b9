// How far ahead of the matrix pipe can ONE wave run?  One wave per SIMD (256 threads per workgroup, 256 workgroups) runs a loop of
// G x v_mfma_f32_16x16x4_f32 (independent accumulators, 32 cycles of the pipe each) followed by a filler of F idle cycles (s_nop)
// or of one LDS-DMA copy.  If the time per iteration stays G x 32 until F ~ 32 (G - 1 MFMAs are not absorbed), MFMA issue blocks
// until the pipe takes the instruction (no queue); if a filler of ~32 (G - 1) cycles is free, the wave queues G MFMAs.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_queue_probe.hip -o build_ab/mfma_queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G, int F, int DMA>
__global__ __launch_bounds__(256) void k(float* out, const float* src, int iters) {
  extern __shared__ char smem[];
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.f;
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x >> 6) * 1024);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
  const unsigned voff = (threadIdx.x & 63) * 16;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
    if (DMA) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rs) : "memory");
    }
#pragma unroll
    for (int f = 0; f < F / 8; ++f) asm volatile("s_nop 7");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s; ((unsigned long long*)out)[1] = t1 - t0; }
}

template <int G, int F, int DMA>
void run(float* d_out, float* d_src) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<G, F, DMA>), dim3(256), dim3(256), 8192, 0, d_out, d_src, iters);
  hipLaunchKernelGGL((k<G, F, DMA>), dim3(256), dim3(256), 8192, 0, d_out, d_src, iters);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
  printf("G=%d filler=%3d dma=%d: %.1f cycles per iteration (MFMAs alone %d)\n", G, F, DMA, (double)h[1] / iters, G * 32);
}

int main() {
  float *d_out, *d_src;
  hipMalloc(&d_out, 64); hipMalloc(&d_src, 1 << 20); hipMemset(d_src, 0, 1 << 20);
  run<1, 0, 0>(d_out, d_src); run<1, 16, 0>(d_out, d_src); run<1, 24, 0>(d_out, d_src); run<1, 32, 0>(d_out, d_src); run<1, 48, 0>(d_out, d_src);
  run<2, 0, 0>(d_out, d_src); run<2, 24, 0>(d_out, d_src); run<2, 48, 0>(d_out, d_src); run<2, 64, 0>(d_out, d_src); run<2, 96, 0>(d_out, d_src);
  run<4, 0, 0>(d_out, d_src); run<4, 48, 0>(d_out, d_src); run<4, 96, 0>(d_out, d_src); run<4, 128, 0>(d_out, d_src); run<4, 160, 0>(d_out, d_src);
  run<8, 0, 0>(d_out, d_src); run<8, 96, 0>(d_out, d_src); run<8, 192, 0>(d_out, d_src); run<8, 256, 0>(d_out, d_src); run<8, 320, 0>(d_out, d_src);
  run<1, 0, 1>(d_out, d_src); run<2, 0, 1>(d_out, d_src); run<4, 0, 1>(d_out, d_src); run<8, 0, 1>(d_out, d_src);
  return 0;
}
