// Probe (diagnostics): do v_cvt_pk_fp8_f32 / v_cvt_f16_f32 / v_cvt_pk_f16_f32 saturate when MODE.FP16_OVFL (bit 23) is set?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, int* out, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");
  float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  _Float16 h = (_Float16)a;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  f2 v; v.x = a; v.y = b;
  h2 hp = __builtin_convertvector(v, h2);
  out[threadIdx.x * 3] = p & 0xffff;
  out[threadIdx.x * 3 + 1] = (int)__builtin_bit_cast(unsigned short, h);
  out[threadIdx.x * 3 + 2] = (int)__builtin_bit_cast(unsigned, hp);
}
int main() {
  float h[128]; float vals[] = {0.f, 1.f, 448.f, 464.f, 480.f, 500.f, 1e6f, -1e6f, 65504.f, 65520.f, 70000.f, -70000.f, 1e30f, -1e30f, 3e-3f, -0.3f};
  for (int i = 0; i < 128; ++i) h[i] = vals[i % 16];
  float* d; int* o; hipMalloc(&d, 512); hipMalloc(&o, 64 * 12);
  hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, ovfl);
    int ho[192]; hipMemcpy(ho, o, 64 * 12, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("FP16_OVFL=%d in %g %g -> fp8 pair 0x%04x  f16(a) 0x%04x  pk_f16 0x%08x\n", ovfl, h[2*i], h[2*i+1], ho[3*i], ho[3*i+1], (unsigned)ho[3*i+2]);
  }
  return 0;
}
