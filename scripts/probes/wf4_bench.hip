// Stand-alone micro-benchmark + self-check of the F(4x4,3x3) Winograd kernel (conv3x3_wf4.h): builds in seconds, for kernel
// iterations.  Diagnostics only (the parity tests of record are tests/test_gpu_parity.py through the C-ABI).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ifisr_amd/csrc -Iinclude scripts/probes/wf4_bench.hip -o scripts/probes/wf4_bench
//   wf4_bench [check]      shapes: WF4_SHAPES="n,h,w,cin,cout,flags,res;..."   flags: 1 relu-on-load, 2 relu, 8 fused x2 bilinear (h, w: the enlarged map)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#define FISR_F4_TRACE 1
#define FISR_F4_SHARE 1      // the V-sharing instantiations (r05: measured, not in the product)
#define FISR_F4X_TRACE 1
#include "conv3x3_wf4.h"
#include "diag/conv3x3_wf4x.h"
using namespace fisr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// direct fp32 reference on the GPU (one thread per output element; fp64 accumulation)
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

// the legacy x2 bilinear (glue_kernels.h upsample2), one thread per output element
__global__ void ref_up2(const float* in, float* out, int N, int H, int W, int C) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * 4 * H * W * C) return;
  const int c = i % C;
  size_t p = i / C;
  const int ox = p % (2 * W); p /= 2 * W;
  const int oy = p % (2 * H);
  const int n = p / (2 * H);
  const int y0 = oy >> 1, x0 = ox >> 1, y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ty = (oy & 1) ? 0.5f : 0.f, tx = (ox & 1) ? 0.5f : 0.f;
  auto at = [&](int y, int x) { return in[((size_t)(n * H + y) * W + x) * C + c]; };
  const float top = at(y0, x0) + (at(y0, x1) - at(y0, x0)) * tx;
  const float bot = at(y1, x0) + (at(y1, x1) - at(y1, x0)) * tx;
  out[i] = top + (bot - top) * ty;
}

// WF4X=1: the one-wave-per-SIMD kernel (conv3x3_wf4x.h)
static bool g_x = false;
static hipError_t launch_any(const ConvArgs& a, hipStream_t st) { return g_x ? launch_conv_wf4x(a, st) : launch_conv_wf4(a, st); }

int main(int argc, char** argv) {
  g_x = getenv("WF4X") && atoi(getenv("WF4X")) != 0;
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  std::string shapes = getenv("WF4_SHAPES") ? getenv("WF4_SHAPES")
                     : check ? "1,40,100,64,64,3,1;2,24,24,64,128,1,0;1,17,45,16,64,0,0"
                             : "12,544,992,64,64,3,1;12,544,992,64,64,0,0;12,272,496,128,128,3,1;12,136,248,256,256,3,1;12,68,124,512,512,3,1;12,136,248,512,256,2,0";
  size_t pos = 0;
  while (pos < shapes.size()) {
    size_t e = shapes.find(';', pos);
    if (e == std::string::npos) e = shapes.size();
    int n, h, w, ci, co, fl, rs, ics = 0, ocs = 0, dil = 1;
    // (optional: pixel stride of the input buffer, of the output buffer (channels), dilation -> the GENERAL instantiation; timing only)
    if (sscanf(shapes.substr(pos, e - pos).c_str(), "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", &n, &h, &w, &ci, &co, &fl, &rs, &ics, &ocs, &dil) < 7) break;
    if (ics < ci) ics = ci;
    if (ocs < co) ocs = co;
    pos = e + 1;
    const bool ups = fl & 8;            // flags bit 3: the input is the half-resolution map (fused x2 bilinear)
    const size_t in_e = (size_t)n * h * w * ics / (ups ? 4 : 1), out_e = (size_t)n * h * w * ocs;
    std::vector<float> hw((size_t)9 * ci * co), hb(co), hin(in_e), hres(rs ? out_e : 0);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 9) % 2001 - 1000) * 1e-3f; };
    for (auto& v : hw) v = rnd() * sqrtf(2.f / (9 * ci)) * 1.7f;
    for (auto& v : hb) v = rnd();
    for (auto& v : hin) v = rnd() * 1.7f;
    for (auto& v : hres) v = rnd();
    std::vector<char> wp;
    pack_weights_wf4(hw.data(), ci, co, ci, wp);
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
    a.C0 = ci; a.C1 = 0; a.N = n; a.H = h; a.W = w; a.Cout = co; a.CoutPad = co;
    a.in0_cs = ics; a.rec_cs = ocs; a.dil = dil;
    a.relu_in = fl & 1; a.relu_out = (fl >> 1) & 1; a.d2s = 0; a.ups = ups;
    // WF4_SHARE=n: the V-sharing instantiations (runs of n N blocks; r05) where the shape takes them; `check` then also compares
    // the output bit for bit with the plain instantiation's.  flags bit 2 (4): fused 2x2 pooling (needs res)
    const int share_env = getenv("WF4_SHARE") ? atoi(getenv("WF4_SHARE")) : 0;
    const int share = share_env >= 2 && !ups && wf4_share_fits(ci, 0, co, share_env) ? share_env : 0;
    void* d_vscr = nullptr;
    float* d_pool = nullptr;
    if (share) { CK(hipMalloc(&d_vscr, wf4_vscr_bytes(ci, 0))); CK(hipMemset(d_vscr, 0xff, wf4_vscr_bytes(ci, 0))); a.vscr = d_vscr; a.share = share; }
    if ((fl & 4) && rs) { CK(hipMalloc(&d_pool, out_e)); a.pool_out = d_pool; }
    const int items = ((w + F4_TW - 1) / F4_TW) * ((h + F4_TH - 1) / F4_TH) * n * (co / F4_BN);
    unsigned long long* d_tr;
    const size_t tr_rows = (size_t)items + 256 * 9;
    CK(hipMalloc(&d_tr, tr_rows * 64)); CK(hipMemset(d_tr, 0, tr_rows * 64));
    if (check) {
      CK(hipMalloc(&d_w, hw.size() * 4)); CK(hipMalloc(&d_ref, out_e * 4));
      CK(hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      CK(launch_any(a, nullptr));
      float* d_full = d_in;
      if (ups) {
        CK(hipMalloc(&d_full, in_e * 4 * 4));
        hipLaunchKernelGGL(ref_up2, dim3((in_e * 4 + 255) / 256), dim3(256), 0, nullptr, d_in, d_full, n, h / 2, w / 2, ci);
      }
      hipLaunchKernelGGL(ref_conv, dim3((out_e + 255) / 256), dim3(256), 0, nullptr, d_full, d_w, d_b, d_res, d_ref, n, h, w, ci, co, fl & 1, (fl >> 1) & 1);
      CK(hipDeviceSynchronize());
      std::vector<float> o(out_e), r(out_e);
      CK(hipMemcpy(o.data(), d_out, out_e * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r.data(), d_ref, out_e * 4, hipMemcpyDeviceToHost));
      double mx = 0, ss = 0; size_t bad = 0;
      for (size_t i = 0; i < out_e; ++i) { double d = fabs((double)o[i] - r[i]); if (!(d <= 1e-3)) ++bad; if (d > mx) mx = d; ss += d * d; }
      printf("check %dx%dx%d %d->%d f%d r%d share %d: max %.3e rms %.3e bad %zu %s\n", n, h, w, ci, co, fl, rs, share, mx, sqrt(ss / out_e), bad, bad ? "FAIL" : "ok");
      if (share) {      // bit-identical to the plain instantiation (same V, same MFMAs in the same order)?
        ConvArgs b = a;
        b.share = 0; b.vscr = nullptr;
        float* d_pool2 = nullptr;
        std::vector<float> pl1, pl2;
        if (d_pool) { pl1.resize(out_e / 4); pl2.resize(out_e / 4); CK(hipMemcpy(pl1.data(), d_pool, out_e, hipMemcpyDeviceToHost)); CK(hipMalloc(&d_pool2, out_e)); b.pool_out = d_pool2; }
        for (int rep_ = 0; rep_ < 3; ++rep_) {
          CK(hipMemset(d_out, 0xff, out_e * 4));
          CK(launch_any(rep_ == 0 ? b : a, nullptr));
          CK(hipDeviceSynchronize());
          std::vector<float> o2(out_e);
          CK(hipMemcpy(o2.data(), d_out, out_e * 4, hipMemcpyDeviceToHost));
          size_t diff = 0, firstd = 0;
          for (size_t i = 0; i < out_e; ++i) if (memcmp(&o2[i], &o[i], 4)) { if (!diff) firstd = i; ++diff; }
          printf("    %s vs first shared launch: %zu of %zu words differ%s\n", rep_ == 0 ? "plain instantiation" : "shared launch again", diff, out_e, diff ? "  FAIL" : "  (bit-identical)");
          if (diff) { size_t i = firstd; const int c = i % co; size_t pp = i / co; const int x = pp % w; pp /= w; printf("      first at [n %zu y %zu x %d c %d] %g vs %g\n", pp / h, pp % h, x, c, o2[i], o[i]); }
          if (rep_ == 0 && d_pool) {
            CK(hipMemcpy(pl2.data(), d_pool2, out_e, hipMemcpyDeviceToHost));
            printf("    pooled map: %s\n", memcmp(pl1.data(), pl2.data(), out_e) ? "DIFFERS  FAIL" : "bit-identical");
          }
        }
        if (d_pool2) CK(hipFree(d_pool2));
      }
      if (bad) {      // which channels / rows / columns are wrong
        std::vector<size_t> bc(co, 0), by(h, 0), bx(w, 0);
        int shown = 0;
        for (size_t i = 0; i < out_e; ++i) {
          if (fabs((double)o[i] - r[i]) <= 1e-3) continue;
          const int c = i % co; size_t pp = i / co; const int x = pp % w; pp /= w; const int y = pp % h;
          ++bc[c]; ++by[y]; ++bx[x];
          if (shown++ < 6) printf("    [n %zu y %d x %d c %d] got %g ref %g\n", pp / h, y, x, c, o[i], r[i]);
        }
        printf("    bad channels:"); for (int c = 0; c < co; ++c) if (bc[c]) printf(" %d:%zu", c, bc[c]); printf("\n");
        printf("    bad rows:"); for (int y = 0; y < h && y < 40; ++y) if (by[y]) printf(" %d:%zu", y, by[y]); printf("\n");
        printf("    bad cols:"); for (int x = 0; x < w && x < 70; ++x) if (bx[x]) printf(" %d:%zu", x, bx[x]); printf("\n");
      }
      CK(hipFree(d_w)); CK(hipFree(d_ref));
      if (ups) CK(hipFree(d_full));
    } else {
      for (int i = 0; i < 2; ++i) CK(launch_any(a, nullptr));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 5;
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) CK(launch_any(a, nullptr));
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      // one traced launch: per-workgroup {start, K-loop end, end, after output transform, after prologue, ...}
      a.trace = d_tr;
      CK(launch_any(a, nullptr));
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> tr(tr_rows * 8);
      CK(hipMemcpy(tr.data(), d_tr, tr_rows * 64, hipMemcpyDeviceToHost));
      // persistent kernel: one row per workgroup {start, first item's K-loop end, end, first item's end, after prologue, real start, real end, items done}
      std::vector<double> life, pro, main_, epi, clk, per_item;
      for (int i = 0; i < items && i < 256; ++i) {
        const unsigned long long* t = &tr[(size_t)i * 8];
        if (t[2] <= t[0] || t[7] == 0) continue;
        life.push_back((double)(t[2] - t[0])); pro.push_back((double)(t[4] - t[0])); main_.push_back((double)(t[1] - t[4])); epi.push_back((double)(t[3] - t[1]));
        per_item.push_back((double)(t[2] - t[0]) / (double)t[7]);
        if (t[6] > t[5]) clk.push_back((double)(t[2] - t[0]) / (double)(t[6] - t[5]) * 100.0);
      }
      auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return v[v.size() / 2]; };
      const int nch = ci / 4;
      {   // second item of each workgroup, per wave: iteration 0, iteration 1, rest of the K loop per chunk, flush + output transform, loads/stores
        const int grid = items < 256 ? items : 256;
        for (int role = 0; role < 2; ++role) {
          std::vector<double> a0, a1, ar, ot, st;
          for (int g = 0; g < grid; ++g)
            for (int w = role * 4; w < role * 4 + 4; ++w) {
              const unsigned long long* t = &tr[((size_t)grid + (size_t)g * 8 + w) * 8];
              if (t[5] <= t[0] || t[0] == 0) continue;
              a0.push_back((double)(t[1] - t[0])); a1.push_back((double)(t[2] - t[1])); ar.push_back((double)(t[3] - t[2]) / (nch - 2));
              ot.push_back((double)(t[4] - t[3])); st.push_back((double)(t[5] - t[4]));
            }
          printf("    %s waves, 2nd item: iteration 0 %.0f  iteration 1 %.0f  later iterations %.0f each  flush + setup %.0f  output stage %.0f\n",
                 role ? "copy" : "transform", med(a0), med(a1), med(ar), med(ot), med(st));
        }
      }
      printf("%2dx%dx%d %3d->%3d f%d r%d: %8.1f us %6.1f TF | items/CU %.1f cycles/item %.0f (MFMA %d) | first item: prologue %.0f K loop %.0f (%.0f per chunk) epilogue %.0f | clk %.0f MHz\n",
             n, h, w, ci, co, fl, rs, us, 2.0 * 9 * ci * co * n * h * w / us / 1e6, items / 256.0, med(per_item), nch * 2304, med(pro), med(main_), med(main_) / nch, med(epi), med(clk));
    }
    CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(d_b)); CK(hipFree(d_wp)); CK(hipFree(d_tr));
    if (d_res) CK(hipFree(d_res));
    if (d_vscr) CK(hipFree(d_vscr));
    if (d_pool) CK(hipFree(d_pool));
  }
  return 0;
}
