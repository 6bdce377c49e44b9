// Probe (diagnostics; r06): does the vector ALU of a SIMD work beside ANOTHER wave's bf16 MFMAs (v_mfma_f32_32x32x16_bf16)?  Waves 0-3 of a
// 512-thread workgroup run a VALU loop (v_pk_add_f32 | v_cvt_pk_bf16_f32 | v_and_b32 | ds_read_b128 | ds_write_b64), waves 4-7 (their SIMD
// partners) an MFMA loop of 8 independent accumulators.  Reported: each alone, both together (sum = no overlap, max = full overlap).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl scripts/probes/mfma_bf16_valu_overlap_probe.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(512) void k(float* out, int nv, int nm, const uint4* in) {
  __shared__ uint4 lds[2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4 ra = in[lane], rb = in[lane + 64];
  lds[threadIdx.x] = ra; lds[threadIdx.x + 512] = rb; lds[threadIdx.x + 1024] = ra; lds[threadIdx.x + 1536] = rb;
  __syncthreads();
  float sum = 0.f;
  if (wave < 4) {
    f2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = f2{__builtin_bit_cast(float, ra.x) + i, __builtin_bit_cast(float, ra.y) - i};
    float y[32];
    for (int i = 0; i < 32; ++i) y[i] = __builtin_bit_cast(float, ra.z) + i;
    const f2 aa = {__builtin_bit_cast(float, rb.x), __builtin_bit_cast(float, rb.y)};
    for (int it = 0; it < nv; ++it) {
      if (OP == 0) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(aa)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(aa)); } }
      if (OP == 1) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { unsigned r; const float u = y[2 * i], v = y[2 * i + 1]; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(u), "v"(v)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(u)); } }
      if (OP == 2) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(y[i])); } }
      if (OP == 3) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { const uint4 r = lds[(threadIdx.x + 64 * i) & 2047]; asm volatile("" :: "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w)); } }
      if (OP == 4) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { *reinterpret_cast<uint2*>(&lds[(threadIdx.x + 64 * i) & 2047]) = make_uint2(ra.x, ra.y); } asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    for (int i = 0; i < 16; ++i) sum += x[i].x + x[i].y + y[i] + y[i + 16];
  } else {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
    for (int it = 0; it < nm; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}
template <int OP> float run(int nv, int nm, float* out, const uint4* in) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, nv, nm, in);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, out, nv, nm, in);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
template <int OP> void both(const char* what, float* out, const uint4* in) {
  const int nv = 20000, nm = 20000;       // 32 VALU / LDS instructions and 8 MFMAs (256 pipe cycles) per iteration
  const float v = run<OP>(nv, 0, out, in), m = run<OP>(0, nm, out, in), b = run<OP>(nv, nm, out, in);
  printf("%-22s alone %7.3f ms (%.1f cycles per instruction at 2.4 GHz)   mfma alone %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n", what, v, v * 2.4e6 / (nv * 32.0), m, b, v + m, v > m ? v : m);
}
int main() {
  float* out; uint4* in; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 4096 * 16);
  (void)hipMemset(in, 0x3c, 4096 * 16);
  both<0>("v_pk_add_f32", out, in);
  both<1>("v_cvt_pk_bf16_f32", out, in);
  both<2>("v_and_b32", out, in);
  both<3>("ds_read_b128", out, in);
  both<4>("ds_write_b64", out, in);
  return 0;
}
