// Probe (diagnostics, not product): DVFS-limited throughput of two MFMA mixes that deliver the same
// 32 channels x (32x32 tile) of "split" products:
//   A: 6 x v_mfma_f32_32x32x16_bf16                      (bf16x3: 3 MFMAs per 16 channels)
//   B: 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8, K=64: both cross terms)
// Operands are random bit patterns held in registers; 4 accumulators per wave; 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void mix(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 r[8];
  for (int i = 0; i < 8; ++i) r[i] = src[(tid * 8 + i) & 0xffff];
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (MODE == 0) {
        bf16x8 a0 = __builtin_bit_cast(bf16x8, r[0]), a1 = __builtin_bit_cast(bf16x8, r[1]);
        bf16x8 b0 = __builtin_bit_cast(bf16x8, r[2 + (t & 1)]), b1 = __builtin_bit_cast(bf16x8, r[4 + (t & 1)]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[t], 0, 0, 0);
      } else {
        f16x8 a0 = __builtin_bit_cast(f16x8, r[0]), a1 = __builtin_bit_cast(f16x8, r[1]);
        f16x8 b0 = __builtin_bit_cast(f16x8, r[2 + (t & 1)]), b1 = __builtin_bit_cast(f16x8, r[4 + (t & 1)]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[t], 0, 0, 0);
        i32x8 fa, fb;
        fa[0] = r[6].x; fa[1] = r[6].y; fa[2] = r[6].z; fa[3] = r[6].w; fa[4] = r[7].x; fa[5] = r[7].y; fa[6] = r[7].z; fa[7] = r[7].w;
        fb[0] = r[2].x; fb[1] = r[3].y; fb[2] = r[4].z; fb[3] = r[5].w; fb[4] = r[3].x; fb[5] = r[2].y; fb[6] = r[5].z; fb[7] = r[4].w;
        // cbsz/blgp = 0 -> fp8 (e4m3) for A and B; scales = 127 (1.0)
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[t], 0, 0, 0, 127, 0, 127);
      }
    }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int k = 0; k < 16; ++k) s += acc[t][k];
  if (s == 123.456f) out[tid] = s;
}

int main() {
  std::vector<uint16_t> h(65536 * 8);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (uint16_t)(0x3800 + ((st >> 12) & 0x7ff)) ^ (uint16_t)((st >> 31) << 15); }
  uint4* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 1 << 24);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 2, iters = 20000;
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(mix<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      else hipLaunchKernelGGL(mix<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // "useful" split-product work: per iteration per wave 4 tiles x 32 channels x 32x32 MACs
      double units = (double)blocks * 4 * iters * 4;   // tile-steps of 32 channels
      printf("mode %d (%s): %.2f ms  -> %.1f G tile-steps(32ch)/s ; equivalent algorithmic %.1f TF/s\n", mode,
             mode == 0 ? "6x bf16 32x32x16" : "2x f16 32x32x16 + 1x fp8 32x32x64 scaled", ms, units / ms / 1e6,
             units * 2.0 * 32 * 32 * 32 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
