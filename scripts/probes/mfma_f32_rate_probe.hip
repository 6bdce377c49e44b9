// Probe: sustained issue rate of v_mfma_f32_32x32x2_f32 (the weight-gradient kernel's instruction).  Nine independent
// accumulators per wave, operands (a) constant registers, (b) a fresh VGPR pair per MFMA read from LDS one step ahead.
// 256 workgroups x 4 waves (one wave per SIMD) or 512 x 4 with two workgroups per CU (two waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const float* in) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = in[i & 255];
  __syncthreads();
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = in[lane], b = in[lane + 64];
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  } else {
    float av[2][NACC], bv[2];
    const float* p = s + lane;
    for (int t = 0; t < NACC; ++t) av[0][t] = p[t * 64];
    bv[0] = p[640];
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* q = s + lane + (((it + h + 1) & 3) << 10);
#pragma unroll
        for (int t = 0; t < NACC; ++t) av[h ^ 1][t] = q[t * 64];
        bv[h ^ 1] = q[640];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[h][t], bv[h], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float sum = 0.f;
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int MODE, int NACC> void run(const char* what, int wgs, float* out, const float* in) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters, in);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters, in);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * NACC * 4096.0;
  printf("%-44s wgs %4d: %8.3f ms  %7.1f TFLOP/s\n", what, wgs, ms, fl / ms / 1e9);
}
int main() {
  float *out, *in; hipMalloc(&out, 512 * 256 * 4); hipMalloc(&in, 4096 * 4);
  hipMemset(in, 0, 4096 * 4);
  run<0, 9>("9 acc, register operands", 256, out, in);
  run<0, 9>("9 acc, register operands", 512, out, in);
  run<0, 4>("4 acc, register operands", 256, out, in);
  run<0, 4>("4 acc, register operands", 512, out, in);
  run<0, 2>("2 acc, register operands", 512, out, in);
  run<1, 9>("9 acc, LDS operands one step ahead", 256, out, in);
  run<1, 9>("9 acc, LDS operands one step ahead", 512, out, in);
  return 0;
}
