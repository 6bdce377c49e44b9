// Probe (diagnostics, not product; r06): cycles per v_mfma_f32_32x32x8_bf16_1k (the K = 8 form CDNA3 had) next to
// v_mfma_f32_32x32x16_bf16, back-to-back DEPENDENT issue on one accumulator next to 8 independent accumulators, and
// v_permlane32_swap.  Question behind it: conv3x3_wino8b.h works in 8-channel chunks -- is K = 8 at half the passes
// (then three K = 8 MFMAs per chunk beat two K = 16 ones), and does a dependent pair on one accumulator stall?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/k8_rate scripts/probes/mfma_bf16_k8_rate_probe.hip && /tmp/k8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(const uint4* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 r[2];
  for (int i = 0; i < 2; ++i) r[i] = src[(tid * 2 + i) & 0xffff];
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
  const bf16x8 a0 = __builtin_bit_cast(bf16x8, r[0]), b0 = __builtin_bit_cast(bf16x8, r[1]);
  const s16x4 a4 = {(short)r[0].x, (short)r[0].y, (short)r[0].z, (short)r[0].w}, b4 = {(short)r[1].x, (short)r[1].y, (short)r[1].z, (short)r[1].w};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[t], 0, 0, 0);
    }
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[t], 0, 0, 0);
    }
    if (MODE == 2) {      // dependent pairs: acc[t] twice in a row
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[t], 0, 0, 0);
      }
    }
    if (MODE == 3) {      // one accumulator, eight in a row
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 8; ++t) for (int k = 0; k < 16; ++k) s += acc[t][k];
  if (s == 123.456f) out[tid] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  std::vector<uint16_t> h(65536 * 8);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3800 + ((st >> 12) & 0x7ff)) ^ (uint16_t)((st >> 31) << 15); }
  uint4* d; float* o; unsigned long long* c;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 1 << 24); hipMalloc(&c, 8 * 4096);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 4000;
  const char* names[] = {"bf16 32x32x16, 8 accumulators", "bf16 32x32x8 (_1k), 8 accumulators", "bf16 32x32x16, dependent pairs", "bf16 32x32x16, one accumulator"};
  const int per_it[] = {8, 8, 8, 8};
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
        if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
        if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
        if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> hc(blocks);
      hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : hc) s += (double)v; s /= blocks;
      printf("%d wave(s)/SIMD  %-40s %8.1f cycles per MFMA of a wave's own stream\n", wps, names[mode], s / (iters * (double)per_it[mode]));
    }
  }
  return 0;
}
