/* A host written in plain C on top of include/fisr.h -- no Python, no torch.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host.c \
 *       -o c_host -L fisr_amd -lfisr_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/fisr_amd -Wl,-rpath,/opt/rocm/lib
 *   ./c_host weights.bin input.bin out_l3.bin [precision]
 *
 * weights.bin : repeated records  { int32 name_len; char name[name_len]; int32 rank; int64 dims[rank];
 *               float data[prod(dims)] }  -- the 276 FISRnet variables under their TF names
 *               (what FISRnet.load restores, FISRnet.py:1101-1115)
 * input.bin   : int32 n, h, w ; float [n,h,w,29]      (the feed of sess.run, FISRnet.py:871)
 * out_l3.bin  : float [n,2h,2w,9]                      (its result)
 *
 * tests/test_gpu_parity.py builds and runs this against the committed golden forward.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fisr.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)
#define FISR_CHECK(call) do { int rc_ = (call); if (rc_ < 0) DIE("%s -> %d: %s", #call, rc_, fisr_last_error(ctx)); } while (0)
#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) DIE("%s -> %s", #call, hipGetErrorString(e_)); } while (0)

int main(int argc, char** argv) {
  if (argc < 4) DIE("usage: %s weights.bin input.bin out_l3.bin [precision 0..3]", argv[0]);
  const int precision = argc > 4 ? atoi(argv[4]) : FISR_PREC_BF16X3;
  fisr_ctx* ctx = NULL;
  FISR_CHECK(fisr_create(&ctx, 0));
  printf("%s\n", fisr_version());

  /* ---- weight seam ---- */
  FILE* f = fopen(argv[1], "rb");
  if (!f) DIE("cannot open %s", argv[1]);
  int32_t name_len;
  int ignored = 0;
  while (fread(&name_len, 4, 1, f) == 1) {
    char name[512];
    int32_t rank;
    int64_t dims[8];
    if (name_len <= 0 || name_len >= (int)sizeof name || fread(name, 1, name_len, f) != (size_t)name_len) DIE("bad record");
    name[name_len] = 0;
    if (fread(&rank, 4, 1, f) != 1 || rank < 1 || rank > 8 || fread(dims, 8, rank, f) != (size_t)rank) DIE("bad record %s", name);
    size_t count = 1;
    for (int i = 0; i < rank; ++i) count *= (size_t)dims[i];
    float* host = (float*)malloc(count * sizeof(float));
    if (!host || fread(host, sizeof(float), count, f) != count) DIE("short record %s", name);
    int rc = fisr_set_weight(ctx, name, host, dims, rank);
    free(host);
    if (rc < 0) DIE("fisr_set_weight(%s) -> %d: %s", name, rc, fisr_last_error(ctx));
    ignored += rc == 1;          /* optimizer slots etc. of a training checkpoint */
  }
  fclose(f);
  printf("variables set: %d (ignored %d)\n", fisr_num_variables_set(ctx), ignored);
  FISR_CHECK(fisr_finalize_weights(ctx, precision));

  /* ---- graph seam ---- */
  f = fopen(argv[2], "rb");
  if (!f) DIE("cannot open %s", argv[2]);
  int32_t nhw[3];
  if (fread(nhw, 4, 3, f) != 3) DIE("bad input header");
  const int n = nhw[0], h = nhw[1], w = nhw[2];
  const size_t in_count = (size_t)n * h * w * 29, out_count = (size_t)n * 2 * h * 2 * w * 9;
  float* in_host = (float*)malloc(in_count * sizeof(float));
  float* out_host = (float*)malloc(out_count * sizeof(float));
  if (!in_host || !out_host || fread(in_host, sizeof(float), in_count, f) != in_count) DIE("short input");
  fclose(f);

  hipStream_t stream;
  float *in_dev, *out_dev;
  void* ws;
  const size_t ws_bytes = fisr_workspace_bytes(ctx, n, h, w);
  if (!ws_bytes) DIE("fisr_workspace_bytes: %s", fisr_last_error(ctx));
  HIP_CHECK(hipStreamCreate(&stream));
  HIP_CHECK(hipMalloc((void**)&in_dev, in_count * sizeof(float)));
  HIP_CHECK(hipMalloc((void**)&out_dev, out_count * sizeof(float)));
  HIP_CHECK(hipMalloc(&ws, ws_bytes));
  HIP_CHECK(hipMemcpyAsync(in_dev, in_host, in_count * sizeof(float), hipMemcpyHostToDevice, stream));
  FISR_CHECK(fisr_forward(ctx, in_dev, n, h, w, out_dev, NULL, NULL, ws, ws_bytes, stream));
  HIP_CHECK(hipMemcpyAsync(out_host, out_dev, out_count * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));

  /* error behaviour: a shape that is not a multiple of 32 is refused, nothing is launched */
  if (fisr_forward(ctx, in_dev, n, h - 1, w, out_dev, NULL, NULL, ws, ws_bytes, stream) != FISR_EINVAL) DIE("expected FISR_EINVAL");

  f = fopen(argv[3], "wb");
  if (!f || fwrite(out_host, sizeof(float), out_count, f) != out_count) DIE("cannot write %s", argv[3]);
  fclose(f);
  double sum = 0;
  for (size_t i = 0; i < out_count; ++i) sum += out_host[i];
  printf("forward [%d,%d,%d,29] -> [%d,%d,%d,9], workspace %.1f MB, mean %.6f\n", n, h, w, n, 2 * h, 2 * w, ws_bytes / 1e6, sum / out_count);

  hipFree(ws); hipFree(out_dev); hipFree(in_dev); hipStreamDestroy(stream);
  fisr_destroy(ctx);
  free(in_host); free(out_host);
  return 0;
}
