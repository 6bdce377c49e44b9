// Probe (diagnostics, not product): cycles per v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands) next to
// v_mfma_f32_32x32x16_f16, 8 independent accumulators per wave, one or two waves per SIMD, s_memtime around the loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mx_rate scripts/probes/mfma_mx_rate_probe.hip && /tmp/mx_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(const uint4* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int iters, int sa, int sb) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 r[8];
  for (int i = 0; i < 8; ++i) r[i] = src[(tid * 8 + i) & 0xffff];
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
  const f16x8 a0 = __builtin_bit_cast(f16x8, r[0]), b0 = __builtin_bit_cast(f16x8, r[1]);
  i32x8 fa, fb;
  fa[0] = r[2].x; fa[1] = r[2].y; fa[2] = r[2].z; fa[3] = r[2].w; fa[4] = r[3].x; fa[5] = r[3].y; fa[6] = r[3].z; fa[7] = r[3].w;
  fb[0] = r[4].x; fb[1] = r[4].y; fb[2] = r[4].z; fb[3] = r[4].w; fb[4] = r[5].x; fb[5] = r[5].y; fb[6] = r[5].z; fb[7] = r[5].w;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[t], 0, 0, 0);
      if (MODE == 1) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[t], 0, 0, 0, sa, 0, sb);
      if (MODE == 2) {   // the kernel's mix per accumulator and chunk: 9 fp16 + 5 fp8
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[t], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[t], 0, 0, 0, sa, 0, sb);
      }
      if (MODE == 3) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[t], 2, 2, 0, sa, 0, sb);      // fp6 operands (cbsz = blgp = 2): the 8-pass form
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
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;
    for (int mode = 0; mode < 6; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters, 127, 127);
        if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters, 127, 127);
        if (mode == 2) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters, 113, 120);
        if (mode == 3) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters / 8, 113, 120);
        if (mode == 4) hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters, 127, 127);
        if (mode == 5) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, d, o, c, iters, 127, 127);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> hc(blocks);
      hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : hc) s += (double)v; s /= blocks;
      const char* names[] = {"f16 32x32x16", "mx fp8 32x32x64, scales 127", "mx fp8 32x32x64, scales 113 / 120", "9 f16 + 5 mx fp8 per accumulator", "mx fp6 32x32x64", "f16 32x32x16 again"};
      const double per = mode == 3 ? s / ((iters / 8) * 8.0) : s / (iters * 8.0);
      printf("%d wave(s)/SIMD  %-36s %8.1f cycles per %s (a wave's own MFMAs; the SIMD runs %d such streams)\n", wps, names[mode], per, mode == 3 ? "accumulator-chunk (ideal 9*32 + 5*64 = 608)" : "MFMA", wps);
    }
  }
  return 0;
}
