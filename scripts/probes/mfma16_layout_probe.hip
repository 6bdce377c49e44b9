// Probe (diagnostics, not product): operand / result register layout of v_mfma_f32_16x16x32_f16 (same as
// _bf16) and v_mfma_f32_16x16x4_f32.  A is one-hot at (lane la, element ea); B holds a unique id per
// (lane, element): D[row][col] then names the B element that shares A's K index in every column.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe16(int la, int ea, float* out) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)((l == la && e == ea) ? 1.f : 0.f); b[e] = (_Float16)(float)(1 + l * 8 + e); }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void probe4(int la, float* out) {
  const int l = threadIdx.x;
  const float a = l == la ? 1.f : 0.f, b = (float)(1 + l);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  std::vector<float> h(256);
  printf("== v_mfma_f32_16x16x32_f16: for A one-hot at (lane, elem): result lanes/regs that are non-zero and the B (lane,elem) they see\n");
  const int tests[][2] = {{0, 0}, {0, 1}, {0, 7}, {1, 0}, {15, 0}, {16, 0}, {16, 3}, {32, 0}, {48, 0}, {63, 7}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("A(l=%2d,e=%d):", t[0], t[1]);
    int shown = 0;
    for (int l = 0; l < 64 && shown < 4; ++l) for (int r = 0; r < 4; ++r) if (h[l * 4 + r] != 0.f) {
      const int id = (int)h[l * 4 + r] - 1;
      if (shown < 4) printf("  D(l=%2d,r=%d)=B(l=%2d,e=%d)", l, r, id / 8, id % 8);
      ++shown;
    }
    int nz = 0; for (float v : h) nz += v != 0.f;
    printf("  [%d non-zero]\n", nz);
  }
  printf("== v_mfma_f32_16x16x4_f32\n");
  const int t4[] = {0, 1, 15, 16, 32, 48, 63};
  for (int la : t4) {
    hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, la, d);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("A(l=%2d):", la);
    int shown = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h[l * 4 + r] != 0.f && shown++ < 4) printf("  D(l=%2d,r=%d)=B(l=%2d)", l, r, (int)h[l * 4 + r] - 1);
    int nz = 0; for (float v : h) nz += v != 0.f;
    printf("  [%d non-zero]\n", nz);
  }
  return 0;
}
