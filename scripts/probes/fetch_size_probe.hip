// What do rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* count for the access pattern of conv3x3_wf4.h's raw copies?
// The F(4x4) kernel fetches a 64-channel fp32 NHWC tensor (256 B per pixel = two 128-byte lines) in "raw pairs": 32 B of a
// pixel per visit (two lanes x 16 B), pixel after pixel, and comes back for the next 32 B of the same lines an iteration later.
// The guide's x2 correction of FETCH_SIZE is calibrated for 16 B / lane streaming reads only.  Known byte counts, one kernel
// per pattern, over a buffer (1.66 GB) far beyond the 256 MB Infinity Cache:
//   stream16      every byte once, 16 B per lane, contiguous                     useful = touched lines = the buffer
//   sector32<k>   32 B (channels 8k .. 8k+7) of every pixel, nothing else        useful = 1/8 of the buffer, lines touched = 1/2
//   sector64<k>   64 B (channels 16k .. 16k+15) of every pixel                   useful = 1/4, lines touched = 1/2
//   visits8       a workgroup walks 612 pixels (one halo tile) 8 times, 32 B of a pixel per pass (the kernel's order without
//                 its arithmetic): useful = the buffer, every line visited 4 times a few microseconds apart
//   visits8_far   the same with 64 tiles per workgroup walked pair-major (a line's 4 visits ~ 5 MB of other traffic apart)
// Run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace ...; ... --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ...;
//            ... --pmc TCC_HIT_sum TCC_MISS_sum ...   (one pass each)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream16(const f32x4* __restrict__ in, size_t n16, float* sink) {
  f32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) a += in[i];
  if (a.x + a.y + a.z + a.w == 1.2345e-30f) *sink = a.x;
}

// LANES lanes x 16 B of every pixel starting at byte `off` of the pixel's 256
template <int LANES>
__global__ __launch_bounds__(256) void sector(const char* __restrict__ in, size_t npix, int off, float* sink) {
  f32x4 a = {0, 0, 0, 0};
  const size_t total = npix * LANES;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t p = i / LANES; const int l = (int)(i % LANES);
    a += *reinterpret_cast<const f32x4*>(in + p * 256 + off + l * 16);
  }
  if (a.x + a.y + a.z + a.w == 1.2345e-30f) *sink = a.x;
}

// a workgroup owns `tiles` runs of 612 consecutive pixels; PAIR_MAJOR = false: per tile, 8 passes of 32 B per pixel;
// true: per pass, all its tiles (the 4 visits of a line are tiles x 19.6 KB x (all workgroups) of traffic apart)
template <bool PAIR_MAJOR>
__global__ __launch_bounds__(512) void visits8(const char* __restrict__ in, size_t npix, int tiles, float* sink) {
  f32x4 a = {0, 0, 0, 0};
  const size_t p0 = (size_t)blockIdx.x * tiles * 612;
  const int lane2 = threadIdx.x & 1, pl = threadIdx.x >> 1;     // 256 pixels per step of 512 threads
  if (!PAIR_MAJOR) {
    for (int t = 0; t < tiles; ++t)
      for (int k = 0; k < 8; ++k)
        for (int q = pl; q < 612; q += 256) {
          const size_t p = p0 + (size_t)t * 612 + q;
          if (p < npix) a += *reinterpret_cast<const f32x4*>(in + p * 256 + k * 32 + lane2 * 16);
        }
  } else {
    for (int k = 0; k < 8; ++k)
      for (int t = 0; t < tiles; ++t)
        for (int q = pl; q < 612; q += 256) {
          const size_t p = p0 + (size_t)t * 612 + q;
          if (p < npix) a += *reinterpret_cast<const f32x4*>(in + p * 256 + k * 32 + lane2 * 16);
        }
  }
  if (a.x + a.y + a.z + a.w == 1.2345e-30f) *sink = a.x;
}

template <typename F> static void timed(const char* name, double useful_gb, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipEventRecord(e0, nullptr));
  launch();
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-14s useful %.3f GB  %.1f us  %.2f TB/s of useful bytes\n", name, useful_gb, ms * 1e3, useful_gb / ms);
}

int main() {
  const size_t npix = (size_t)12 * 544 * 992, bytes = npix * 256;
  char* d; float* sink;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d, 0, bytes));
  const double gb = bytes / 1e9;
  timed("stream16", gb, [&] { hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, nullptr, (const f32x4*)d, bytes / 16, sink); });
  timed("sector32_k0", gb / 8, [&] { hipLaunchKernelGGL(sector<2>, dim3(256 * 16), dim3(256), 0, nullptr, d, npix, 0, sink); });
  timed("sector32_k5", gb / 8, [&] { hipLaunchKernelGGL(sector<2>, dim3(256 * 16), dim3(256), 0, nullptr, d, npix, 160, sink); });
  timed("sector64_k1", gb / 4, [&] { hipLaunchKernelGGL(sector<4>, dim3(256 * 16), dim3(256), 0, nullptr, d, npix, 64, sink); });
  timed("line128", gb / 2, [&] { hipLaunchKernelGGL(sector<8>, dim3(256 * 16), dim3(256), 0, nullptr, d, npix, 128, sink); });
  {
    const int tiles = 1; const unsigned grid = (unsigned)((npix + 611) / 612);
    timed("visits8", gb, [&] { hipLaunchKernelGGL(visits8<false>, dim3(grid), dim3(512), 0, nullptr, d, npix, tiles, sink); });
  }
  {
    const int tiles = 41; const unsigned grid = (unsigned)((npix + (size_t)612 * tiles - 1) / ((size_t)612 * tiles));     // 258 workgroups: one round
    timed("visits8_far", gb, [&] { hipLaunchKernelGGL(visits8<true>, dim3(grid), dim3(512), 0, nullptr, d, npix, tiles, sink); });
  }
  CK(hipFree(d)); CK(hipFree(sink));
  return 0;
}
