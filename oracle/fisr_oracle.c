/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY, never the product path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Nothing under fisr_amd/ links, loads or calls it.
 *
 * Restates the reference's FISRnet forward (FISRnet.py:73-173 over the blocks of
 * ops.py:7-76) in plain C with OpenMP, in float (`fisr_oracle_forward_f32`) and
 * double (`fisr_oracle_forward_f64`).  Each function cites the reference file:line
 * it follows (relative to /root/reference).  The TF-1.13 op semantics behind those
 * call sites (conv2d SAME, max_pool, legacy resize_images, depth_to_space) are not
 * in the reference tree and TensorFlow cannot be installed here: PARITY UNPINNED
 * for them (see oracle/fisr_oracle.py header and DESIGN.md); this C file is
 * cross-checked against the numpy twin and torch-CPU in tests/test_oracle.py.
 *
 * Weight blob layout: for each of the 138 convs in `fisr_amd.weights.conv_specs()`
 * order: w[3][3][Ci][Co] (HWIO) then b[Co], float32, densely packed.
 *
 * This translation unit includes itself twice to instantiate both precisions.
 */
#ifndef FISR_ORACLE_BODY
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FISR_ORACLE_BODY
#define REAL float
#define SUF(x) x##_f32
#include "fisr_oracle.c"
#undef REAL
#undef SUF
#define REAL double
#define SUF(x) x##_f64
#include "fisr_oracle.c"
#undef REAL
#undef SUF

int fisr_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void fisr_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

#else /* ------------------------------ templated body ------------------------------ */

typedef struct { REAL* p; int n, h, w, c; } SUF(T);

static SUF(T) SUF(talloc)(int n, int h, int w, int c) {
  SUF(T) t; t.n = n; t.h = h; t.w = w; t.c = c;
  t.p = (REAL*)malloc(sizeof(REAL) * (size_t)n * h * w * c);
  return t;
}
static void SUF(tfree)(SUF(T)* t) { free(t->p); t->p = 0; }

/* ops.py:7-11 Conv2d: tf.nn.conv2d(x, w[3,3,Ci,Co], strides 1, 'SAME') + b.
 * Cross-correlation, zero padding 1.  relu_in applies ops.py:17-18
 * relu() to the input first (the callers' `Conv2d(relu(x), ...)`, ops.py:41-42). */
static SUF(T) SUF(conv)(SUF(T) x, const float* wt, const float* bias, int co, int relu_in) {
  const int ci = x.c, H = x.h, W = x.w;
  SUF(T) y = SUF(talloc)(x.n, H, W, co);
  REAL* wr = (REAL*)malloc(sizeof(REAL) * 9 * (size_t)ci * co);
  for (size_t i = 0; i < 9 * (size_t)ci * co; ++i) wr[i] = (REAL)wt[i];
  const int nxb = (W + 3) / 4;
#pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < x.n; ++n)
    for (int yy = 0; yy < H; ++yy) {
      for (int xb = 0; xb < nxb; ++xb) {
        REAL acc[4][512];
        const int x0 = xb * 4;
        const int nb = (W - x0) < 4 ? (W - x0) : 4;
        for (int q = 0; q < nb; ++q)
          for (int o = 0; o < co; ++o) acc[q][o] = (REAL)bias[o];
        for (int dy = 0; dy < 3; ++dy) {
          const int sy = yy + dy - 1;
          if (sy < 0 || sy >= H) continue;
          for (int dx = 0; dx < 3; ++dx) {
            const REAL* wk = wr + (size_t)(dy * 3 + dx) * ci * co;
            for (int q = 0; q < nb; ++q) {
              const int sx = x0 + q + dx - 1;
              if (sx < 0 || sx >= W) continue;
              const REAL* xp = x.p + (((size_t)n * H + sy) * W + sx) * ci;
              REAL* a = acc[q];
              for (int i = 0; i < ci; ++i) {
                REAL xv = xp[i];
                if (relu_in && xv < 0) xv = 0;
                const REAL* wrow = wk + (size_t)i * co;
                for (int o = 0; o < co; ++o) a[o] += xv * wrow[o];
              }
            }
          }
        }
        for (int q = 0; q < nb; ++q) {
          REAL* yp = y.p + (((size_t)n * H + yy) * W + x0 + q) * co;
          for (int o = 0; o < co; ++o) yp[o] = acc[q][o];
        }
      }
    }
  free(wr);
  return y;
}

static void SUF(relu_)(SUF(T) t) { /* ops.py:17-18, in place */
  size_t m = (size_t)t.n * t.h * t.w * t.c;
  for (size_t i = 0; i < m; ++i) if (t.p[i] < 0) t.p[i] = 0;
}
static void SUF(add_)(SUF(T) a, SUF(T) b) { /* ops.py:43 `n = x + n`, result in a */
  size_t m = (size_t)a.n * a.h * a.w * a.c;
  for (size_t i = 0; i < m; ++i) a.p[i] = b.p[i] + a.p[i];
}

/* ops.py:54 tf.nn.max_pool 2x2/2 'SAME' on even sizes. */
static SUF(T) SUF(pool)(SUF(T) x) {
  SUF(T) y = SUF(talloc)(x.n, x.h / 2, x.w / 2, x.c);
  for (int n = 0; n < x.n; ++n)
    for (int h = 0; h < y.h; ++h)
      for (int w = 0; w < y.w; ++w)
        for (int c = 0; c < x.c; ++c) {
          const REAL* p = x.p + (((size_t)n * x.h + 2 * h) * x.w + 2 * w) * x.c + c;
          REAL m = p[0];
          if (p[x.c] > m) m = p[x.c];
          if (p[(size_t)x.w * x.c] > m) m = p[(size_t)x.w * x.c];
          if (p[(size_t)x.w * x.c + x.c] > m) m = p[(size_t)x.w * x.c + x.c];
          y.p[(((size_t)n * y.h + h) * y.w + w) * x.c + c] = m;
        }
  return y;
}

/* ops.py:69 tf.image.resize_images(BILINEAR) x2, TF-1.13 legacy kernel
 * (SURVEY App. B.3): in = out*0.5; evaluation order top/bot/then y. */
static SUF(T) SUF(up2)(SUF(T) x) {
  SUF(T) y = SUF(talloc)(x.n, x.h * 2, x.w * 2, x.c);
  for (int n = 0; n < x.n; ++n)
    for (int oy = 0; oy < y.h; ++oy) {
      const int y0 = oy >> 1, y1 = (y0 + 1 < x.h) ? y0 + 1 : x.h - 1;
      const REAL ty = (REAL)(oy & 1) * (REAL)0.5;
      for (int ox = 0; ox < y.w; ++ox) {
        const int x0 = ox >> 1, x1 = (x0 + 1 < x.w) ? x0 + 1 : x.w - 1;
        const REAL tx = (REAL)(ox & 1) * (REAL)0.5;
        const REAL* tl = x.p + (((size_t)n * x.h + y0) * x.w + x0) * x.c;
        const REAL* tr = x.p + (((size_t)n * x.h + y0) * x.w + x1) * x.c;
        const REAL* bl = x.p + (((size_t)n * x.h + y1) * x.w + x0) * x.c;
        const REAL* br = x.p + (((size_t)n * x.h + y1) * x.w + x1) * x.c;
        REAL* o = y.p + (((size_t)n * y.h + oy) * y.w + ox) * x.c;
        for (int c = 0; c < x.c; ++c) {
          const REAL top = tl[c] + (tr[c] - tl[c]) * tx;
          const REAL bot = bl[c] + (br[c] - bl[c]) * tx;
          o[c] = top + (bot - top) * ty;
        }
      }
    }
  return y;
}

/* FISRnet.py:99 tf.depth_to_space(x,2): out[n,2h+i,2w+j,c] = x[n,h,w,(2i+j)*C+c]. */
static SUF(T) SUF(d2s)(SUF(T) x) {
  const int C = x.c / 4;
  SUF(T) y = SUF(talloc)(x.n, x.h * 2, x.w * 2, C);
  for (int n = 0; n < x.n; ++n)
    for (int h = 0; h < x.h; ++h)
      for (int w = 0; w < x.w; ++w)
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j)
            memcpy(y.p + (((size_t)n * y.h + 2 * h + i) * y.w + 2 * w + j) * C,
                   x.p + (((size_t)n * x.h + h) * x.w + w) * x.c + (2 * i + j) * C, sizeof(REAL) * C);
  return y;
}

/* tf.concat axis 3 (ops.py:71, FISRnet.py:113,144). */
static SUF(T) SUF(cat)(SUF(T) a, SUF(T) b) {
  SUF(T) y = SUF(talloc)(a.n, a.h, a.w, a.c + b.c);
  size_t px = (size_t)a.n * a.h * a.w;
  for (size_t i = 0; i < px; ++i) {
    memcpy(y.p + i * y.c, a.p + i * a.c, sizeof(REAL) * a.c);
    memcpy(y.p + i * y.c + a.c, b.p + i * b.c, sizeof(REAL) * b.c);
  }
  return y;
}

/* FISRnet.py:81,112 legacy BICUBIC resize at integer factor s == x[:, ::s, ::s, :]
 * (SURVEY App. B.2). */
static SUF(T) SUF(sub)(SUF(T) x, int s) {
  SUF(T) y = SUF(talloc)(x.n, x.h / s, x.w / s, x.c);
  for (int n = 0; n < x.n; ++n)
    for (int h = 0; h < y.h; ++h)
      for (int w = 0; w < y.w; ++w)
        memcpy(y.p + (((size_t)n * y.h + h) * y.w + w) * x.c,
               x.p + (((size_t)n * x.h + h * s) * x.w + w * s) * x.c, sizeof(REAL) * x.c);
  return y;
}

typedef struct { const float* blob; size_t off; } SUF(WCur);

/* next conv in blob order */
static SUF(T) SUF(cnext)(SUF(WCur)* wc, SUF(T) x, int co, int relu_in) {
  const float* w = wc->blob + wc->off;
  const float* b = w + (size_t)9 * x.c * co;
  wc->off += (size_t)9 * x.c * co + co;
  return SUF(conv)(x, w, b, co, relu_in);
}

/* ops.py:39-44 res_block: x + conv1(relu(conv0(relu(x)))).  Consumes x. */
static SUF(T) SUF(rb)(SUF(WCur)* wc, SUF(T) x) {
  SUF(T) a = SUF(cnext)(wc, x, x.c, 1);
  SUF(T) b = SUF(cnext)(wc, a, x.c, 1);
  SUF(tfree)(&a);
  SUF(add_)(b, x);
  SUF(tfree)(&x);
  return b;
}

/* One U-Net + heads (FISRnet.py:83-108); x is consumed; pred [N,2h,2w,9]. */
static SUF(T) SUF(level)(SUF(WCur)* wc, SUF(T) x) {
  SUF(T) skip[3], n = x, t;
  const int chs[3] = {64, 128, 256};
  for (int l = 0; l < 3; ++l) {            /* ops.py:48-55 Enc_level_res */
    t = SUF(cnext)(wc, n, chs[l], 0); SUF(tfree)(&n);
    t = SUF(rb)(wc, t);
    t = SUF(rb)(wc, t);
    SUF(relu_)(t);
    skip[l] = t;
    n = SUF(pool)(t);
  }
  t = SUF(cnext)(wc, n, 512, 0); SUF(tfree)(&n);   /* ops.py:59-63 Bottleneck_res */
  t = SUF(rb)(wc, t);
  SUF(relu_)(t);
  n = t;
  for (int l = 2; l >= 0; --l) {           /* ops.py:67-76 Dec_level_res */
    SUF(T) u = SUF(up2)(n); SUF(tfree)(&n);
    t = SUF(cnext)(wc, u, chs[l], 0); SUF(tfree)(&u);
    SUF(relu_)(t);
    u = SUF(cat)(t, skip[l]); SUF(tfree)(&t); SUF(tfree)(&skip[l]);
    t = SUF(cnext)(wc, u, chs[l], 0); SUF(tfree)(&u);
    t = SUF(rb)(wc, t);
    t = SUF(rb)(wc, t);
    SUF(relu_)(t);
    n = t;
  }
  SUF(T) outs[2];
  const int hc[2] = {6, 3};
  for (int hd = 0; hd < 2; ++hd) {         /* FISRnet.py:95-106 */
    SUF(T) a = SUF(cnext)(wc, n, 64, 0);
    a = SUF(rb)(wc, a);
    t = SUF(cnext)(wc, a, 256, 1); SUF(tfree)(&a);
    SUF(relu_)(t);
    a = SUF(d2s)(t); SUF(tfree)(&t);
    outs[hd] = SUF(cnext)(wc, a, hc[hd], 0); /* relu(relu(x)) == relu(x): FISRnet.py:100 */
    SUF(tfree)(&a);
  }
  SUF(tfree)(&n);
  SUF(T) p = SUF(talloc)(outs[0].n, outs[0].h, outs[0].w, 9);  /* FISRnet.py:107-108 */
  size_t px = (size_t)p.n * p.h * p.w;
  for (size_t i = 0; i < px; ++i) {
    REAL* o = p.p + i * 9;
    const REAL* f = outs[0].p + i * 6;
    const REAL* s = outs[1].p + i * 3;
    o[0] = f[0]; o[1] = f[1]; o[2] = f[2];
    o[3] = s[0]; o[4] = s[1]; o[5] = s[2];
    o[6] = f[3]; o[7] = f[4]; o[8] = f[5];
  }
  SUF(tfree)(&outs[0]); SUF(tfree)(&outs[1]);
  return p;
}

/* FISRnet.py:73-173 model(img[N,H,W,29], sf=2) -> pred_l1, pred_l2, pred_l3.
 * Returns 0, or -1 on bad arguments.  Any of l1/l2 may be NULL. */
int SUF(fisr_oracle_forward)(const float* in, int n, int h, int w, const float* blob,
                             REAL* l1, REAL* l2, REAL* l3) {
  if (!in || !blob || !l3 || n < 1 || h < 32 || w < 32 || (h % 32) || (w % 32)) return -1;
  SUF(T) img = SUF(talloc)(n, h, w, 29);
  for (size_t i = 0; i < (size_t)n * h * w * 29; ++i) img.p[i] = (REAL)in[i];
  SUF(WCur) wc = {blob, 0};
  SUF(T) p1 = SUF(level)(&wc, SUF(sub)(img, 4));
  SUF(T) s2 = SUF(sub)(img, 2);
  SUF(T) x2 = SUF(cat)(s2, p1); SUF(tfree)(&s2);
  SUF(T) p2 = SUF(level)(&wc, x2);
  SUF(T) x3 = SUF(cat)(img, p2);
  SUF(T) p3 = SUF(level)(&wc, x3);
  if (l1) memcpy(l1, p1.p, sizeof(REAL) * (size_t)p1.n * p1.h * p1.w * 9);
  if (l2) memcpy(l2, p2.p, sizeof(REAL) * (size_t)p2.n * p2.h * p2.w * 9);
  memcpy(l3, p3.p, sizeof(REAL) * (size_t)p3.n * p3.h * p3.w * 9);
  SUF(tfree)(&p1); SUF(tfree)(&p2); SUF(tfree)(&p3); SUF(tfree)(&img);
  return 0;
}

/* Single conv entry point for op-level parity tests (ops.py:7-11 +- relu, residual). */
int SUF(fisr_oracle_conv3x3)(const REAL* x, int n, int h, int w, int ci, const float* wt,
                             const float* bias, int co, int relu_in, REAL* y) {
  SUF(T) t; t.p = (REAL*)x; t.n = n; t.h = h; t.w = w; t.c = ci;
  SUF(T) o = SUF(conv)(t, wt, bias, co, relu_in);
  memcpy(y, o.p, sizeof(REAL) * (size_t)n * h * w * co);
  SUF(tfree)(&o);
  return 0;
}
#endif
