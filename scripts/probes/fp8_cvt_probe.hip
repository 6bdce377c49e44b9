#include <hip/hip_runtime.h>
__global__ void k(const float* in, int* out, float* dec) {
  float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  out[threadIdx.x] = p;
  dec[threadIdx.x * 2] = __builtin_amdgcn_cvt_f32_fp8(p, 0);
  dec[threadIdx.x * 2 + 1] = __builtin_amdgcn_cvt_f32_fp8(p, 1);
}
int main() {
  float h[128]; float vals[] = {0.f, 1.f, -1.f, 0.5f, 448.f, 500.f, 1e6f, -1e6f, 0.001f, 0.0019f, 0.002f, 1.0625f, 1.1875f, 17.f, 3e-3f, -0.3f};
  for (int i = 0; i < 128; ++i) h[i] = vals[i % 16];
  float* d; int* o; float* dd; hipMalloc(&d, 512); hipMalloc(&o, 256); hipMalloc(&dd, 512);
  hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, dd);
  int ho[64]; float hd[128]; hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("in %g %g -> 0x%04x -> %g %g\n", h[2*i], h[2*i+1], ho[i] & 0xffff, hd[2*i], hd[2*i+1]);
  return 0;
}
