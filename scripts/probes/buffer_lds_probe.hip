// Probe: does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose voffset is out of range, and is the
// scalar offset excluded from the range check?  (conv3x3_dma.h relies on both for the zero padding of its halo tiles.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* g, float* out, unsigned nbytes, unsigned soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 512; i += 64) ((float*)smem)[i] = -7.f;       // poison
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;          // odd lanes: out of range
  if (threadIdx.x == 62) off = nbytes - 16;                                   // last valid 16 bytes (+ soff goes past num_records)
  unsigned lds = (unsigned)(size_t)(lds_ptr_t)smem;
  unsigned keep;
  asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[off], %[rs], %[so] offen lds\n\ts_mov_b32 m0, %[keep]"
               : [keep] "=&s"(keep) : [off] "v"(off), [rs] "s"(rs), [lds] "s"(lds), [so] "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  const unsigned nbytes = 2048 * 4, soff = 64;     // buffer = first 2048 floats; soffset 64 bytes = 16 floats
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, nbytes, soff);
  std::vector<float> r(256);
  hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
  // lane l writes LDS floats 4l..4l+3.  even lane l: expects g[4l + 16 ..]; odd: zeros (if OOB writes zeros) or -7 (if skipped)
  printf("lane0: %g %g %g %g (expect 16 17 18 19)\n", r[0], r[1], r[2], r[3]);
  printf("lane1 (OOB): %g %g %g %g (0 = zeros written, -7 = write skipped)\n", r[4], r[5], r[6], r[7]);
  printf("lane2: %g (expect %d)\n", r[8], 8 + 16);
  printf("lane62 (voff = nbytes-16, + soffset past the end): %g (expect %d if soffset is outside the range check, 0 if inside)\n", r[248], 2048 - 4 + 16);
  return 0;
}
