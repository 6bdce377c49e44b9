// Probe (diagnostics, not product): power-limited throughput and shader clock of pure-MFMA loops in
// different instruction shapes on random register operands:
//   0: v_mfma_f32_32x32x16_bf16   1: v_mfma_f32_16x16x32_bf16   2: v_mfma_f32_32x32x16_f16
//   3: v_mfma_f32_16x16x32_f16    4: v_mfma_f32_32x32x2_f32     5: 32x32x16 bf16 with the B operand all zero
// 2 waves/SIMD, 64 accumulator registers per wave, no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void loop(const uint4* __restrict__ src, float* __restrict__ out, int iters,
                                               unsigned long long* clk) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 r[4];
  for (int i = 0; i < 4; ++i) r[i] = src[(tid * 4 + i) & 0xffff];
  if (MODE == 5) { r[2] = make_uint4(0, 0, 0, 0); r[3] = r[2]; }
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (MODE == 0 || MODE == 5)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r[t & 1]), __builtin_bit_cast(bf16x8, r[2 + (t >> 1)]), acc[t], 0, 0, 0);
          else if (MODE == 2)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, r[t & 1]), __builtin_bit_cast(f16x8, r[2 + (t >> 1)]), acc[t], 0, 0, 0);
          else {
            const f32x4 a = __builtin_bit_cast(f32x4, r[t & 1]), b = __builtin_bit_cast(f32x4, r[2 + (t >> 1)]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
          }
        }
    }
    for (int t = 0; t < 4; ++t) for (int k = 0; k < 16; ++k) s += acc[t][k];
  } else {
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) for (int k = 0; k < 4; ++k) acc[t][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (MODE == 1)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r[t & 1]), __builtin_bit_cast(bf16x8, r[2 + ((t >> 1) & 1)]), acc[t], 0, 0, 0);
        else
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, r[t & 1]), __builtin_bit_cast(f16x8, r[2 + ((t >> 1) & 1)]), acc[t], 0, 0, 0);
      }
    }
    for (int t = 0; t < 16; ++t) for (int k = 0; k < 4; ++k) s += acc[t][k];
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
  if (s == 123.456f) out[tid] = s;
}

int main() {
  std::vector<uint16_t> h(65536 * 8);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3800 + ((st >> 12) & 0x7ff)) ^ (uint16_t)((st >> 31) << 15); }
  uint4* d; float* o; unsigned long long* clk;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 1 << 24); hipMalloc(&clk, 512 * 16);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512;
  const char* names[] = {"bf16 32x32x16", "bf16 16x16x32", "f16 32x32x16", "f16 16x16x32", "f32 32x32x2", "bf16 32x32x16, B = 0"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      const int iters = mode == 4 ? 8000 : 16000;
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters, clk); break;
        case 1: hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters, clk); break;
        case 2: hipLaunchKernelGGL(loop<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters, clk); break;
        case 3: hipLaunchKernelGGL(loop<3>, dim3(blocks), dim3(256), 0, 0, d, o, iters, clk); break;
        case 4: hipLaunchKernelGGL(loop<4>, dim3(blocks), dim3(256), 0, 0, d, o, iters, clk); break;
        default: hipLaunchKernelGGL(loop<5>, dim3(blocks), dim3(256), 0, 0, d, o, iters, clk); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> c(blocks * 2);
      hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
      double mhz = 0; for (int b = 0; b < blocks; ++b) mhz += 100.0 * c[2 * b] / (double)c[2 * b + 1]; mhz /= blocks;
      // per iteration per wave: 8 x 32x32x16 (or 16 x 16x16x32 = half of that; or 8 x 32x32x2)
      const double flop_it = mode == 4 ? 8.0 * 2 * 32 * 32 * 2 : (mode == 1 || mode == 3 ? 16.0 * 2 * 16 * 16 * 32 : 8.0 * 2 * 32 * 32 * 16);
      const double tf = (double)blocks * 4 * iters * flop_it / (ms * 1e-3) / 1e12;
      printf("%-22s %8.2f ms  %7.1f TF/s issued  shader clock %5.0f MHz  -> %.1f %% of the MFMA rate at that clock\n", names[mode], ms, tf, mhz,
             100.0 * tf / ((mode == 4 ? 157.3 : 2500.0) * mhz / 2400.0));
    }
  return 0;
}
