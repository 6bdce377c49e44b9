#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// waves 0-3: VALU loop (nv iterations of 32 independent v_fma), waves 4-7: MFMA loop (nm iterations of 36 mfma 16x16x4)
template <int PK> __global__ __launch_bounds__(512) void k(float* out, int nv, int nm, const float* in) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a = in[lane], b = in[lane + 64];
  float sum = 0.f;
  if (wave < 4) {
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = a + i;
    for (int it = 0; it < nv; ++it) {
      if (PK == 0) { _Pragma("unroll") for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); }
      else if (PK == 1) { _Pragma("unroll") for (int i = 0; i < 32; i += 2) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {x[i], x[i+1]}; f2 aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb)); x[i] = v.x; x[i+1] = v.y; } }
      else { _Pragma("unroll") for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
    }
    for (int i = 0; i < 32; ++i) sum += x[i];
  } else {
    f32x4 acc[36];
    for (int t = 0; t < 36; ++t) acc[t] = f32x4{0,0,0,0};
    for (int it = 0; it < nm; ++it) {
#pragma unroll
      for (int t = 0; t < 36; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 36; ++t) for (int r = 0; r < 4; ++r) sum += acc[t][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}
template <int PK> void run(const char* what, int nv, int nm, float* out, const float* in) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<PK>, dim3(256), dim3(512), 0, 0, out, nv, nm, in);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<PK>, dim3(256), dim3(512), 0, 0, out, nv, nm, in);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s nv %5d nm %5d: %8.3f ms  (valu instr/wave %d, mfma/wave %d)\n", what, nv, nm, ms, nv * 32, nm * 36);
}
int main() {
  float *out, *in; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 4096 * 4);
  (void)hipMemset(in, 0, 4096 * 4);
  run<0>("mfma only (1 wave/SIMD)", 0, 4000, out, in);
  run<0>("v_fma only", 20000, 0, out, in);
  run<1>("v_pk_fma only (same instr count)", 20000, 0, out, in);
  run<2>("v_add only", 20000, 0, out, in);
  run<1>("both: pk + mfma", 20000, 4000, out, in);
  return 0;
}
