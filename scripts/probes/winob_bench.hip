// Stand-alone micro-benchmark + self-check + per-wave phase timers of the split-bf16 Winograd kernel (conv3x3_wino8b.h, r06): builds
// in seconds, for kernel iterations.  Diagnostics only (the parity tests of record: tests/test_gpu_winob.py through the C-ABI).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFISR_WB_TRACE -Ifisr_amd/csrc -Iinclude scripts/probes/winob_bench.hip -o scripts/probes/winob_bench
//   winob_bench [check]      shapes: WB_SHAPES="n,h,w,cin,cout,flags,res;..."   flags: 1 relu-on-load, 2 relu, 4 depth_to_space
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "diag/conv3x3_wino8b.h"
using namespace fisr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// direct reference on the GPU (one thread per output element; fp64 accumulation)
__global__ void ref_conv(const float* in, const float* w, const float* b, const float* res, float* out, int N, int H, int W, int Ci, int Co,
                         int relu_in, int relu_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * H * W * Co) return;
  const int co = i % Co;
  size_t p = i / Co;
  const int x = p % W; p /= W;
  const int y = p % H;
  const int n = p / H;
  double s = b[co];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) {
      const int yy = y + a - 1, xx = x + c - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const float* ip = in + ((size_t)(n * H + yy) * W + xx) * Ci;
      const float* wp = w + (size_t)(a * 3 + c) * Ci * Co + co;
      for (int k = 0; k < Ci; ++k) { float v = ip[k]; if (relu_in) v = fmaxf(v, 0.f); s += (double)v * wp[(size_t)k * Co]; }
    }
  if (res) s += res[i];
  if (relu_out) s = s > 0 ? s : 0;
  out[i] = (float)s;
}

int main(int argc, char** argv) {
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  std::string shapes = getenv("WB_SHAPES") ? getenv("WB_SHAPES")
                     : check ? "1,40,100,64,64,3,1;2,24,24,64,128,1,0;1,17,45,32,64,0,0;2,136,248,64,64,1,0"
                             : "12,544,992,64,64,3,0;12,544,992,64,64,0,1;12,272,496,128,128,3,0;12,544,992,64,256,3,0;12,544,992,128,64,2,0";
  size_t pos = 0;
  while (pos < shapes.size()) {
    size_t e = shapes.find(';', pos);
    if (e == std::string::npos) e = shapes.size();
    int n, h, w, ci, co, fl, rs;
    if (sscanf(shapes.substr(pos, e - pos).c_str(), "%d,%d,%d,%d,%d,%d,%d", &n, &h, &w, &ci, &co, &fl, &rs) < 7) break;
    pos = e + 1;
    const size_t in_e = (size_t)n * h * w * ci, out_e = (size_t)n * h * w * co;
    std::vector<float> hw((size_t)9 * ci * co), hb(co), hin(in_e), hres(rs ? out_e : 0);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hw) v = rnd() * sqrtf(2.f / (9 * ci)) * 1.7f;
    for (auto& v : hb) v = rnd();
    for (auto& v : hin) v = rnd() * 1.7f;
    for (auto& v : hres) v = rnd();
    std::vector<char> wp;
    pack_weights_winob(hw.data(), ci, co, ci, wp);
    float *d_in, *d_out, *d_res = nullptr, *d_b, *d_w, *d_ref = nullptr;
    void* d_wp;
    CK(hipMalloc(&d_in, in_e * 4)); CK(hipMalloc(&d_out, out_e * 4)); CK(hipMalloc(&d_b, co * 4)); CK(hipMalloc(&d_wp, wp.size()));
    CK(hipMemcpy(d_in, hin.data(), in_e * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, hb.data(), co * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wp, wp.data(), wp.size(), hipMemcpyHostToDevice));
    if (rs) { CK(hipMalloc(&d_res, out_e * 4)); CK(hipMemcpy(d_res, hres.data(), out_e * 4, hipMemcpyHostToDevice)); }
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.in0 = d_in; a.wpk = d_wp; a.bias = d_b; a.res = d_res; a.out = d_out;
    a.C0 = ci; a.C1 = 0; a.N = n; a.H = h; a.W = w; a.Cout = co; a.CoutPad = (co + W_BN - 1) / W_BN * W_BN;
    a.in0_cs = ci; a.rec_cs = co; a.dil = 1;
    a.relu_in = fl & 1; a.relu_out = (fl >> 1) & 1; a.d2s = (fl >> 2) & 1;
    if (a.d2s) { int k = 0; while ((1 << (k + 1)) <= co / 4) ++k; a.d2s_shift = k; }
    const int items = ((w + TILE_W - 1) / TILE_W) * ((h + TILE_H - 1) / TILE_H) * n * (a.CoutPad / W_BN);
    const int grid = std::min(items, 256);
    unsigned long long* d_tr;
    CK(hipMalloc(&d_tr, (size_t)grid * 8 * 64)); CK(hipMemset(d_tr, 0, (size_t)grid * 8 * 64));
    if (check) {
      CK(hipMalloc(&d_w, hw.size() * 4)); CK(hipMalloc(&d_ref, out_e * 4));
      CK(hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemset(d_out, 0xff, out_e * 4));
      CK(launch_conv_winob(a, nullptr));
      hipLaunchKernelGGL(ref_conv, dim3((out_e + 255) / 256), dim3(256), 0, nullptr, d_in, d_w, d_b, d_res, d_ref, n, h, w, ci, co, fl & 1, (fl >> 1) & 1);
      CK(hipDeviceSynchronize());
      std::vector<float> o(out_e), r(out_e);
      CK(hipMemcpy(o.data(), d_out, out_e * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r.data(), d_ref, out_e * 4, hipMemcpyDeviceToHost));
      double mx = 0, ss = 0; size_t bad = 0;
      for (size_t i = 0; i < out_e; ++i) { double d = fabs((double)o[i] - r[i]); if (!(d <= 2e-4)) ++bad; if (d > mx) mx = d; ss += d * d; }
      printf("check %dx%dx%d %d->%d f%d r%d: max %.3e rms %.3e bad %zu %s\n", n, h, w, ci, co, fl, rs, mx, sqrt(ss / out_e), bad, bad ? "FAIL" : "ok");
      if (bad) {
        int shown = 0;
        for (size_t i = 0; i < out_e && shown < 8; ++i) {
          if (fabs((double)o[i] - r[i]) <= 2e-4) continue;
          const int c = i % co; size_t pp = i / co; const int x = pp % w; pp /= w; const int y = pp % h;
          printf("    [n %zu y %d x %d c %d] got %g ref %g\n", pp / h, y, x, c, o[i], r[i]);
          ++shown;
        }
      }
      CK(hipFree(d_w)); CK(hipFree(d_ref));
    } else {
      for (int i = 0; i < 2; ++i) CK(launch_conv_winob(a, nullptr));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 5;
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) CK(launch_conv_winob(a, nullptr));
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("%2dx%dx%d %3d->%3d f%d r%d: %8.1f us %6.1f TF algorithmic | items/CU %.1f\n", n, h, w, ci, co, fl, rs, us, 2.0 * 9 * ci * co * n * h * w / us / 1e6, items / 256.0);
#ifdef FISR_WB_TRACE
      a.trace = d_tr;
      CK(launch_conv_winob(a, nullptr));
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> tr((size_t)grid * 64);
      CK(hipMemcpy(tr.data(), d_tr, (size_t)grid * 8 * 64, hipMemcpyDeviceToHost));
      auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return v[v.size() / 2]; };
      for (int role = 0; role < 2; ++role) {
        std::vector<double> p0, p1, p2, ep, life, its;
        for (int g = 0; g < grid; ++g)
          for (int wv = role * 4; wv < role * 4 + 4; ++wv) {
            const unsigned long long* t = &tr[((size_t)g * 8 + wv) * 8];
            if (!t[3]) continue;
            const double it = (double)t[3], nitem = it / (ci / 8);
            p0.push_back(t[0] / it); p1.push_back(t[1] / it); p2.push_back(t[2] / it); ep.push_back(t[4] / nitem); life.push_back((double)t[5] / nitem); its.push_back(nitem);
          }
        printf("    %-9s waves, per chunk: %s %.0f  %s %.0f  barrier wait %.0f  | epilogue per item %.0f | life per item %.0f (%.0f items)\n", role ? "copy" : "transform",
               role ? "copies + fix" : "MFMA phase", med(p0), role ? "MFMA phase" : "transform", med(p1), med(p2), med(ep), med(life), med(its));
      }
#endif
    }
    CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(d_b)); CK(hipFree(d_wp)); CK(hipFree(d_tr));
    if (d_res) CK(hipFree(d_res));
  }
  return 0;
}
