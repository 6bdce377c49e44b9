// C-ABI of the training ops (include/fisr.h, "training graph"): SURVEY.md 8 row f4.  Included at the end of
// fisr_api.hip.  The reference builds its training graph in Python on TensorFlow ops and lets tf.gradients / Adam do the
// rest (FISRnet.py:175-497); here the host side (fisr_amd/train.py) keeps the same structure -- a Python tape over these
// entry points -- and every one of them is a HIP kernel of train_kernels.h or the inference engine's direct fp32 conv.
#pragma once
#include "train_kernels.h"

extern "C" {

/* bytes of the packed weights of a [3,3,ci,co] conv for fisr_train_conv3x3 (transpose != 0: of its data-gradient conv,
 * which has co input and ci output channels) */
size_t fisr_train_packed_bytes(int ci, int co, int transpose) {
  const int ci_ = transpose ? co : ci, co_ = transpose ? ci : co;
  const int nt = nt_for<float>(co_);
  const int cout_pad = nt == 0 ? 16 : round_up(co_, 32 * nt);
  return (size_t)(round_up(ci_, 16) / 16) * 9 * cout_pad * CHUNK_BYTES;
}

int fisr_train_pack(const float* d_w_hwio, int ci, int co, int transpose, void* d_packed, void* stream) {
  if (!d_w_hwio || !d_packed || ci <= 0 || co <= 0) return fail(nullptr, FISR_EINVAL, "fisr_train_pack: bad argument");
  const int ci_ = transpose ? co : ci, co_ = transpose ? ci : co;
  const int nt = nt_for<float>(co_);
  const int cout_pad = nt == 0 ? 16 : round_up(co_, 32 * nt), cin_pad = round_up(ci_, 16);
  DeviceGuard guard(device_of(d_packed));
  HIP_OK(nullptr, guard.err);
  const size_t total = (size_t)(cin_pad / 16) * 9 * cout_pad * 16;
  hipLaunchKernelGGL(train_pack_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, d_w_hwio, ci, co, transpose ? 1 : 0,
                     ci_, co_, cin_pad, cout_pad, (float*)d_packed);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

/* Winograd slabs of the same conv for the persistent fp32 Winograd kernel (conv3x3_wino8p.h): bytes (0 = this conv is
 * not eligible: it needs 64 | padded output channels and >= 32 input channels) and the device-side packing. */
size_t fisr_train_wino_bytes(int ci, int co, int transpose) {
  const int ci_ = transpose ? co : ci, co_ = transpose ? ci : co;
  const int cin_pad = round_up(ci_, 16), cout_pad = round_up(co_, 16);
  if (cout_pad % W_BN || cin_pad < 32) return 0;
  return (size_t)(cin_pad / W_CH) * (cout_pad / W_BN) * W_SLAB;
}

int fisr_train_pack_wino(const float* d_w_hwio, int ci, int co, int transpose, void* d_packed, void* stream) {
  if (!d_w_hwio || !d_packed || !fisr_train_wino_bytes(ci, co, transpose)) return fail(nullptr, FISR_EINVAL, "fisr_train_pack_wino: bad argument");
  const int ci_ = transpose ? co : ci, co_ = transpose ? ci : co;
  const int cin_pad = round_up(ci_, 16), nb = round_up(co_, 16) / W_BN;
  DeviceGuard guard(device_of(d_packed));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(train_pack_wino_kernel, dim3(grid_for((size_t)cin_pad * nb * 64)), dim3(256), 0, (hipStream_t)stream, d_w_hwio, ci, co,
                     transpose ? 1 : 0, ci_, co_, cin_pad, nb, (char*)d_packed);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

/* All layouts of n convs in one launch: d_descs is a DEVICE array of n fisr_train_pack_desc (null destinations are skipped). */
int fisr_train_pack_all(const fisr_train_pack_desc* d_descs, int n, void* stream) {
  static_assert(sizeof(fisr_train_pack_desc) == sizeof(PackDesc), "descriptor layout");
  if (!d_descs || n <= 0 || n > 16383) return fail(nullptr, FISR_EINVAL, "fisr_train_pack_all: bad argument");
  DeviceGuard guard(device_of(d_descs));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(train_pack_all_kernel, dim3(96, 4 * n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const PackDesc*>(d_descs));
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

/* out = [relu]( conv3x3([relu](cat(in0, in1))) + bias [+ res] ), weights packed by fisr_train_pack (device).
 * d_packed_wino (nullable): the slabs of fisr_train_pack_wino; every dense layer they exist for then runs on the Winograd
 * kernel (2.25 x fewer MFMAs; measured faster than the direct kernel at every map size of the training step, 3 x 3 pixels
 * included); d_packed may then be NULL (the call fails if it turns out to need the direct kernel).  d_bias must be readable up to the N block's padding (cout rounded up to 64, 16 for the heads; what lies beyond
 * cout is never used).  c0 + c1 is the padded channel count (multiple of 16).  out_cstride == 0: dense [n,h,w,cout] records (or the depth_to_space layout
 * with FISR_CONV_D2S); != 0: channel n is stored at n + out_coff + (n >= out_split ? out_gap : 0) of a pixel stride
 * out_cstride (the [fr1, SR, fr2] scatter of the heads, FISRnet.py:107-108). */
int fisr_train_conv3x3(const float* in0, int c0, const float* in1, int c1, const void* d_packed, const float* d_bias, int cout,
                       const float* res, float* out, int n, int h, int w, int flags, int out_cstride, int out_coff, int out_split,
                       int out_gap, const void* d_packed_wino, void* stream) {
  if (!in0 || (!d_packed && !d_packed_wino) || !d_bias || !out || n <= 0 || h <= 0 || w <= 0 || c0 <= 0 || c1 < 0 || (c0 % 16) || (c1 % 16) || (c1 && !in1))
    return fail(nullptr, FISR_EINVAL, "fisr_train_conv3x3: bad argument");
  const int nt = nt_for<float>(cout);
  const bool scatter = out_cstride != 0;
  if (nt == 0 && !scatter) return fail(nullptr, FISR_EINVAL, "fisr_train_conv3x3: fewer than 16 output channels need the scatter store");
  if (!scatter && (cout % 16)) return fail(nullptr, FISR_EINVAL, "fisr_train_conv3x3: dense output needs cout % 16 == 0");
  if ((flags & FISR_CONV_D2S) && (scatter || cout % 4 || cout / 4 < 16)) return fail(nullptr, FISR_EINVAL, "fisr_train_conv3x3: bad d2s");
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  ConvArgs a;
  a.in0 = in0; a.in1 = in1; a.wpk = d_packed; a.bias = d_bias; a.res = res; a.out = out;
  a.C0 = c0; a.C1 = c1; a.N = n; a.H = h; a.W = w; a.Cout = cout; a.CoutPad = nt == 0 ? 16 : round_up(cout, 32 * nt);
  a.relu_in = (flags & FISR_CONV_RELU_IN) != 0; a.relu_out = (flags & FISR_CONV_RELU_OUT) != 0;
  a.d2s = (flags & FISR_CONV_D2S) != 0; a.d2s_shift = a.d2s ? ilog2(cout / 4) : 0;
  a.out_cstride = scatter ? out_cstride : cout; a.out_coff = scatter ? out_coff : 0;
  a.out_split = scatter ? out_split : 1 << 30; a.out_gap = scatter ? out_gap : 0;
  a.wexp = 0; a.trace = nullptr;
  a.in0_cs = c0; a.in1_cs = c1; a.rec_cs = cout; a.rec_co = 0; a.slope = 0.f; a.dil = 1;
  const long items = (long)((w + TILE_W - 1) / TILE_W) * ((h + TILE_H - 1) / TILE_H) * n * (cout / W_BN);
  long min_items = 1;
#ifdef FISR_DIAG
  static const long env_items = [] { const char* e = getenv("FISR_TRAIN_WINO_ITEMS"); return e ? atol(e) : 0L; }();
  if (env_items > 0) min_items = env_items;
#endif
  if (d_packed_wino && !scatter && cout % W_BN == 0 && (c0 + c1) / W_CH >= 4 && items >= min_items && wino_fits(1, h, w, c0, c1, cout)) {
    a.wpk = d_packed_wino;
    a.CoutPad = cout;
    // The Winograd kernel addresses a launch's whole batch with 32-bit byte offsets: a batch that does not fit goes out in
    // chunks of images (train.py no longer allocates the direct kernel's pack for these layers, so a large batch or
    // patch used to fail the step with EINVAL instead of falling back; fisr_pwc.h splits the same way).
    int per = n;
    while (per > 1 && !wino_fits(per, h, w, c0, c1, cout)) per = (per + 1) / 2;
    const size_t px = (size_t)h * w;
    const size_t out_px = a.d2s ? px * 4 * (size_t)(cout / 4) : px * (size_t)cout;
    for (int k = 0; k < n; k += per) {
      ConvArgs b = a;
      b.N = std::min(per, n - k);
      b.in0 = in0 + (size_t)k * px * c0;
      b.in1 = in1 ? in1 + (size_t)k * px * c1 : nullptr;
      b.res = res ? res + (size_t)k * px * cout : nullptr;
      b.out = out + (size_t)k * out_px;
      HIP_OK(nullptr, launch_conv_wino(b, (hipStream_t)stream));
    }
    return 0;
  }
  if (!d_packed) return fail(nullptr, FISR_EINVAL, "fisr_train_conv3x3: this call needs the direct kernel's packed weights (d_packed)");
  HIP_OK(nullptr, launch_conv<float>(a, nt, scatter, (hipStream_t)stream));
  return 0;
}

}  // extern "C"

static int wgrad_impl(const float* x0, int c0, const float* x1, int c1, int relu_in, const float* g, int cg, float* dw, float* db,
                      int ci, int co, int n, int h, int w, void* stream, unsigned long long* trace) {
  if (!x0 || !g || !dw || c0 <= 0 || c1 < 0 || (c0 % 4) || (c1 % 4) || (cg % 4) || ci > c0 + c1 || co > cg || (c1 && !x1) || n <= 0)
    return fail(nullptr, FISR_EINVAL, "fisr_train_wgrad: bad argument");
  DeviceGuard guard(device_of(dw));
  HIP_OK(nullptr, guard.err);
  if (co <= 8 && c1 == 0 && c0 == 64 && cg == 16) {   // the heads: vector-ALU kernel, one lane per input channel
    WgradArgs a;
    a.x0 = x0; a.x1 = nullptr; a.C0 = c0; a.C1 = 0; a.g = g; a.Cg = cg; a.dw = dw; a.db = db; a.ci = ci; a.co = co;
    a.N = n; a.H = h; a.W = w; a.relu_in = relu_in; a.trace = nullptr;
    const int nci = (ci + 63) / 64, units = n * h;
    a.ksplit = std::max(1, std::min((units + 3) / 4, 768 / nci));
    if (co <= 3) hipLaunchKernelGGL(train_wgrad_head_kernel<3>, dim3(nci * a.ksplit), dim3(256), 0, (hipStream_t)stream, a);
    else if (co <= 6) hipLaunchKernelGGL(train_wgrad_head_kernel<6>, dim3(nci * a.ksplit), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(train_wgrad_head_kernel<8>, dim3(nci * a.ksplit), dim3(256), 0, (hipStream_t)stream, a);
    HIP_OK(nullptr, hipGetLastError());
    return 0;
  }
  WgradTile tl = wgrad_tile(h, w);
  void (*kern)(const WgradArgs) = nullptr;
  int slot = 0;
  // Winograd-domain kernel (2.25 x fewer MFMAs) where one of its tiles (4 x 32, 8 x 16, 16 x 8 pixels) wastes at most a quarter
  // of the map's pixels; the direct kernel with the tile geometry of wgrad_tile() otherwise
  bool wino = false;
  if (ci >= 32 && co >= 32) {
    double best = 0.0;
    static const WgradTile wcand[] = {{32, 4}, {16, 8}, {8, 16}, {8, 8}};            // larger tiles first: ties go to them
    for (const WgradTile& c : wcand) {
      const int tw = c.tw, th = c.th;
      const double eff = ((double)w / (((w + tw - 1) / tw) * tw)) * ((double)h / (((h + th - 1) / th) * th));
      if (eff > best + 1e-9) { best = eff; tl.tw = tw; tl.th = th; }
    }
    wino = best >= 0.75;
    if (!wino) tl = wgrad_tile(h, w);
  }
#ifdef FISR_DIAG
  static const int env_wg = [] { const char* e = getenv("FISR_TRAIN_WGRAD_WINO"); return e ? atoi(e) : -1; }();
  if (env_wg == 0 && wino) { wino = false; tl = wgrad_tile(h, w); }
#endif
#define FISR_WGRAD_CASE(TW, TH, S) if (tl.tw == TW && tl.th == TH) { kern = train_wgrad_kernel<TW, TH>; slot = S; }
  if (wino) {
    if (tl.tw == 32) { kern = train_wgrad_wino_kernel<32, 4>; slot = 6; }
    else if (tl.tw == 16) { kern = train_wgrad_wino_kernel<16, 8>; slot = 7; }
    else if (tl.th == 16) { kern = train_wgrad_wino_kernel<8, 16>; slot = 8; }
    else { kern = train_wgrad_wino_kernel<8, 8>; slot = 9; }
  } else {
  FISR_WGRAD_CASE(32, 4, 0) FISR_WGRAD_CASE(16, 8, 1) FISR_WGRAD_CASE(16, 4, 2)
  FISR_WGRAD_CASE(8, 16, 3) FISR_WGRAD_CASE(8, 8, 4) FISR_WGRAD_CASE(8, 4, 5)
  }
#undef FISR_WGRAD_CASE
  if (!kern) return fail(nullptr, FISR_EINVAL, "fisr_train_wgrad: no kernel for the tile");
  static bool attr_done[64][10] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_done[dev][slot]) {
    HIP_OK(nullptr, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)wgrad_lds_bytes()));
    attr_done[dev][slot] = true;
  }
  WgradArgs a;
  a.x0 = x0; a.x1 = x1; a.C0 = c0; a.C1 = c1; a.g = g; a.Cg = cg; a.dw = dw; a.db = db; a.ci = ci; a.co = co;
  a.N = n; a.H = h; a.W = w; a.relu_in = relu_in; a.trace = trace;
  const int blocks = ((ci + 31) / 32) * ((co + 31) / 32);
  const int ntiles = ((w + tl.tw - 1) / tl.tw) * ((h + tl.th - 1) / tl.th) * n;
  a.ksplit = std::max(1, std::min(ntiles, (512 + blocks - 1) / blocks));       // about two workgroups per CU in total (each ends
                                                                                // with 9216 atomics: few, long-running workgroups)
  hipLaunchKernelGGL(kern, dim3(blocks * a.ksplit), dim3(256), wgrad_lds_bytes(), (hipStream_t)stream, a);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}
extern "C" {
/* dw[3][3][ci][co] += sum_pixels [relu](cat(x0, x1))[p + tap][ci] * g[p][co], and (db != NULL) db[co] += sum_pixels g[p][co];
 * c0 + c1 >= ci (padded channels are read and dropped), cg >= co, c0 % 4 == c1 % 4 == cg % 4 == 0 */
int fisr_train_wgrad(const float* x0, int c0, const float* x1, int c1, int relu_in, const float* g, int cg, float* dw, float* db,
                     int ci, int co, int n, int h, int w, void* stream) {
  return wgrad_impl(x0, c0, x1, c1, relu_in, g, cg, dw, db, ci, co, n, h, w, stream, nullptr);
}
#ifdef FISR_DIAG
/* diagnostics builds: the same launch with a per-workgroup cycle trace (8 words per workgroup, device memory, at least
 * 1024 workgroups' worth) */
FISR_API int fisr_diag_wgrad_trace(const float* x0, int c0, const float* x1, int c1, int relu_in, const float* g, int cg, float* dw, float* db,
                          int ci, int co, int n, int h, int w, void* stream, unsigned long long* d_trace) {
  return wgrad_impl(x0, c0, x1, c1, relu_in, g, cg, dw, db, ci, co, n, h, w, stream, d_trace);
}
#endif

int fisr_train_bgrad(const float* g, int cg, size_t npix, float* db, int co, void* stream) {
  if (!g || !db || cg <= 0 || co > cg) return fail(nullptr, FISR_EINVAL, "fisr_train_bgrad: bad argument");
  DeviceGuard guard(device_of(db));
  HIP_OK(nullptr, guard.err);
  const int gx = (int)std::min<size_t>(1024, (npix + 3) / 4);
  hipLaunchKernelGGL(train_bgrad_kernel, dim3(std::max(gx, 1), (cg + 63) / 64), dim3(256), 0, (hipStream_t)stream, g, cg, npix, db, co);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

#define FISR_TRAIN_EW(NAME, CHECK, LAUNCH)                                                        \
  if (CHECK) return fail(nullptr, FISR_EINVAL, NAME ": bad argument");                           \
  { DeviceGuard guard(device_of(GUARD_PTR)); HIP_OK(nullptr, guard.err); LAUNCH; HIP_OK(nullptr, hipGetLastError()); } \
  return 0;

int fisr_train_relu_bwd(const float* g_in, const float* ref, float* g_out, size_t count, void* stream) {
#define GUARD_PTR g_out
  FISR_TRAIN_EW("fisr_train_relu_bwd", !g_in || !ref || !g_out || (count % 4),
                hipLaunchKernelGGL(train_relu_bwd_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream, g_in, ref, g_out, count / 4))
#undef GUARD_PTR
}

int fisr_train_axpy(const float* x, float a, float* y, size_t count, void* stream) {
#define GUARD_PTR y
  FISR_TRAIN_EW("fisr_train_axpy", !x || !y || (count % 4),
                hipLaunchKernelGGL(train_axpy_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream, x, a, y, count / 4))
#undef GUARD_PTR
}

int fisr_train_maxpool2_bwd(const float* x, const float* dpool, float* dx, int n, int h, int w, int c, void* stream) {
#define GUARD_PTR dx
  FISR_TRAIN_EW("fisr_train_maxpool2_bwd", !x || !dpool || !dx || (h % 2) || (w % 2) || (c % 4),
                hipLaunchKernelGGL(train_maxpool2_bwd_kernel, dim3(grid_for((size_t)n * (h / 2) * (w / 2) * (c / 4))), dim3(256), 0,
                                   (hipStream_t)stream, x, dpool, dx, n, h, w, c))
#undef GUARD_PTR
}

/* dy [n, 2h, 2w, c] -> dx [n, h, w, c] */
int fisr_train_upsample2_bwd(const float* dy, float* dx, int n, int h, int w, int c, void* stream) {
#define GUARD_PTR dx
  FISR_TRAIN_EW("fisr_train_upsample2_bwd", !dy || !dx || (c % 4),
                hipLaunchKernelGGL(train_upsample2_bwd_kernel, dim3(grid_for((size_t)n * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream,
                                   dy, dx, n, h, w, c))
#undef GUARD_PTR
}

/* g [n, 2h, 2w, c] -> out [n, h, w, 4c] */
int fisr_train_s2d(const float* g, float* out, int n, int h, int w, int c, void* stream) {
#define GUARD_PTR out
  FISR_TRAIN_EW("fisr_train_s2d", !g || !out || (c % 4),
                hipLaunchKernelGGL(train_s2d_kernel, dim3(grid_for((size_t)n * h * w * c)), dim3(256), 0, (hipStream_t)stream, g, out, n, h, w, c))
#undef GUARD_PTR
}

int fisr_train_copy_channels(const float* src, int scs, int sco, float* dst, int dcs, int dco, int nc, size_t npix, int add, void* stream) {
#define GUARD_PTR dst
  FISR_TRAIN_EW("fisr_train_copy_channels", !src || !dst || nc <= 0 || sco + nc > scs || dco + nc > dcs,
                hipLaunchKernelGGL(train_copy_channels_kernel, dim3(grid_for(npix * nc)), dim3(256), 0, (hipStream_t)stream, src, scs, sco, dst,
                                   dcs, dco, nc, npix, add))
#undef GUARD_PTR
}

/* One level of the loss (FISRnet.py:316-484).  pred[0..2]: the stride-1 windows' predictions, pred[3]: the stride-2 window's,
 * each [b,h,w,9]; gt [b,h,w,21]; grad[k] receives d(total_loss)/d(pred[k]); sums[7] (device, zeroed by the caller)
 * accumulates the raw squared sums of recn, tm, tmm, td, recn_ss2, td_ss2, tm_ss2; k[7] = lambda * level scale * 2 / n of
 * the seven terms in that order. */
int fisr_train_loss(const float* const* pred4, const float* gt, float* const* grad4, float* sums, size_t npix, const float* k7, void* stream) {
  if (!pred4 || !gt || !grad4 || !sums || !k7) return fail(nullptr, FISR_EINVAL, "fisr_train_loss: bad argument");
  LossArgs a;
  for (int i = 0; i < 4; ++i) {
    if (!pred4[i] || !grad4[i]) return fail(nullptr, FISR_EINVAL, "fisr_train_loss: null tensor");
    a.pred[i] = pred4[i]; a.grad[i] = grad4[i];
  }
  a.gt = gt; a.sums = sums; a.npix = npix;
  a.k_recn = k7[0]; a.k_tm = k7[1]; a.k_tmm = k7[2]; a.k_td = k7[3]; a.k_recn2 = k7[4]; a.k_td2 = k7[5]; a.k_tm2 = k7[6];
  DeviceGuard guard(device_of(sums));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(train_loss_kernel, dim3(std::min(grid_for(npix * 3), 1024)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

int fisr_train_adam(float* w, const float* g, float* m, float* v, size_t count, float lr_t, float b1, float b2, float eps, void* stream) {
#define GUARD_PTR w
  FISR_TRAIN_EW("fisr_train_adam", !w || !g || !m || !v,
                hipLaunchKernelGGL(train_adam_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, count, lr_t, b1, b2, eps))
#undef GUARD_PTR
}
#undef FISR_TRAIN_EW

}  // extern "C"
