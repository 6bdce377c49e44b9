#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const float* in) {
  const int lane = threadIdx.x & 63;
  float a = in[lane], b = in[lane + 64];
  float sum = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0,0,0,0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 4; ++r) sum += acc[t][r];
  } else {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int SHAPE, int NACC> void run(const char* what, int wgs, float* out, const float* in, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters, in);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters, in);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * NACC * (SHAPE == 16 ? 2048.0 : 4096.0);
  printf("%-30s wgs %4d: %8.3f ms  %7.1f TFLOP/s\n", what, wgs, ms, fl / ms / 1e9);
}
int main() {
  float *out, *in; hipMalloc(&out, 512 * 256 * 4); hipMalloc(&in, 4096 * 4);
  hipMemset(in, 0, 4096 * 4);
  run<16, 36>("16x16x4 36 acc", 256, out, in, 4000);
  run<16, 36>("16x16x4 36 acc", 512, out, in, 4000);
  run<16, 8>("16x16x4 8 acc", 512, out, in, 16000);
  run<32, 8>("32x32x2 8 acc", 256, out, in, 8000);
  run<32, 8>("32x32x2 8 acc", 512, out, in, 8000);
  return 0;
}
