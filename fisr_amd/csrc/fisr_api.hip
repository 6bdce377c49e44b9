// C-ABI of libfisr_hip.so (see include/fisr.h): context, weight re-packing, the FISRnet
// forward as a schedule of hand-written gfx950 kernels, glue kernels, op-level entries.
//
// Forward schedule follows FISRnet.model (reference FISRnet.py:73-173) over the blocks of
// ops.py:39-76; every launch below names the reference line it implements.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/fisr.h"
// FISR_DIAG builds (scripts/gpu_*.sh, lib.build(diag=True) -> build_ab/libfisr_hip_diag.so) add what the shipped library
// must not have: environment switches that change which kernel runs (FISR_WINO_VARIANT, FISR_CONV_MR, FISR_HEAD_VALU,
// FISR_CONV_HEAD16), the two superseded Winograd kernels, the compile-time ablations (-DFISR_ABL / -DFISR_WABL) and the
// per-workgroup timeline trace of fisr_diag_bench_conv.  The shipped .so reads no environment variable at all.
#ifndef FISR_DIAG
#undef FISR_ABL
#undef FISR_WABL
#undef FISR_GABL
#undef FISR_RAW_AUX_MODE
#endif
#include "conv3x3.h"
#include "conv3x3_wino8p.h"
#include "conv3x3_wf4.h"
#include "conv3x3_dma.h"
#include "conv3x3_dma_fs.h"
#ifdef FISR_DIAG
#include "diag/conv3x3_wino4.h"
#include "diag/conv3x3_wino8.h"
#endif
#include "head_conv.h"
#ifdef FISR_DIAG
#include "diag/head_conv_strip.h"      // the strip-walking LDS-DMA heads (r05): measured, not adopted; FISR_HEAD_STRIP=1 runs them
#endif
#include "glue_kernels.h"

using namespace fisr;

namespace {

thread_local std::string g_err;

struct ConvW {
  int ci = 0, co = 0;
  std::vector<float> w, b;  // host copies in TF layout (HWIO, [Co])
  bool have_w = false, have_b = false;
  // device, packed
  void* d_w = nullptr;
  float* d_b = nullptr;
  int cin_pad = 0, cout_pad = 0, nt = 2;
  bool rows16 = false;    // fp32, Cout == 16: take the 16-row variant too (PWC-Net's level-1 features fill its rows exactly; NT = 1 computes 32)
  int wexp = 0;  // f16f8: power-of-two pre-scale of the fp8 weight parts
  void* d_wu = nullptr;   // FISR_PREC_F32W: U = G g G^T in the Winograd kernel's LDS image (conv3x3_wino_common.h), else NULL
  void* d_wu4 = nullptr;  // FISR_PREC_F32W4: U = G g G^T of F(4x4,3x3) in the LDS image of conv3x3_wf4.h, else NULL
  float* d_wh = nullptr;  // FISR_PREC_F32W, Cout <= 6: [9][cin_pad][4 | 6] for the vector-ALU head kernel (head_conv.h), else NULL
  void* d_wd = nullptr;   // FISR_PREC_F16, Cout > 32: the weight slabs of the LDS-DMA kernel (conv3x3_dma.h); FISR_PREC_F16F8, Cout % 64 == 0: of
                          // the persistent one (conv3x3_dma_fs.h); else NULL
  int cout_pad_d = 0;     // ... and its Cout padded to the 64-channel block
  int prec = -1;          // the precision the device copies are packed for (differs per layer in FISR_PREC_MIXED)
};

struct ProfEntry {
  hipEvent_t a, b;
  int cls;
  double flops, bytes;
};

}  // namespace

struct fisr_ctx {
  int dev = 0;
  int precision = -1;
  bool wino = false;      // FISR_PREC_F32W / F32W4: eligible convs run the Winograd F(2x2,3x3) kernel
  bool wf4 = false;       // FISR_PREC_F32W4: ... and those conv3x3_wf4.h takes (wf4_wins) the F(4x4,3x3) kernel
  bool finalized = false;
  std::map<std::string, ConvW> convs;  // keyed by conv name (without /w, /b)
  std::string err;
  // profiling
  bool prof = false;
  int prof_mode = 0;  // 1: per kernel class, 2: per layer (name + shape)
  std::vector<std::string> prof_names;
  std::vector<ProfEntry> prof_entries;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<double> prof_ms, prof_flops, prof_bytes;
  std::vector<int64_t> prof_launches;
  double* d_scalar = nullptr;
};

namespace {

int fail(fisr_ctx* ctx, int code, const std::string& msg) {
  g_err = msg;
  if (ctx) ctx->err = msg;
  return code;
}

#define HIP_OK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, FISR_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Every entry point that launches work selects the device the work belongs to and restores the caller's
// current device on return (a host may drive several GPUs, one ctx each, from one thread; torch's notion of
// the current device must not change behind its back).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
// Device that owns a device pointer (glue entry points have no ctx: the output tensor decides); -1 if unknown.
inline int device_of(const void* p) {
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged ? a.device : -1;
}
inline int ilog2(int v) { int k = 0; while ((1 << (k + 1)) <= v) ++k; return k; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int grid_for(size_t work, int block = 256) {
  size_t g = (work + block - 1) / block;
  return (int)std::min<size_t>(std::max<size_t>(g, 1), 256 * 16);
}

// The 138 conv names with their channel counts (reference ops.py:48-76, FISRnet.py:83-106;
// same order as fisr_amd/weights.py conv_specs()).
struct Spec { std::string name; int ci, co; };
std::vector<Spec> all_specs() {
  std::vector<Spec> out;
  for (int lv = 1; lv <= 3; ++lv) {
    const std::string P = "FISRnet/level_" + std::to_string(lv);
    const int cin = lv == 1 ? 29 : 38;
    auto rb = [&](const std::string& p, int c, int i) {
      out.push_back({p + "/res_block/" + std::to_string(i) + "/conv/0", c, c});
      out.push_back({p + "/res_block/" + std::to_string(i) + "/conv/1", c, c});
    };
    const int ec[3][2] = {{cin, 64}, {64, 128}, {128, 256}};
    for (int l = 0; l < 3; ++l) {
      const std::string p = P + "/enc/level_" + std::to_string(l);
      out.push_back({p + "/conv/0", ec[l][0], ec[l][1]});
      rb(p, ec[l][1], 0);
      rb(p, ec[l][1], 1);
    }
    out.push_back({P + "/bottleneck/conv/0", 256, 512});
    rb(P + "/bottleneck", 512, 0);
    const int dc[3][3] = {{2, 512, 256}, {1, 256, 128}, {0, 128, 64}};
    for (auto& d : dc) {
      const std::string p = P + "/dec/level_" + std::to_string(d[0]);
      out.push_back({p + "/resize", d[1], d[2]});
      out.push_back({p + "/conv/0", d[2] * 2, d[2]});
      rb(p, d[2], 0);
      rb(p, d[2], 1);
    }
    const char* heads[2] = {"FI-SR", "SR"};
    const int hco[2] = {6, 3};
    for (int h = 0; h < 2; ++h) {
      const std::string p = P + "/" + heads[h];
      out.push_back({p + "/conv/0", 64, 64});
      rb(p, 64, 0);
      out.push_back({p + "/conv/1", 64, 256});
      out.push_back({p + "/conv/2", 64, hco[h]});
    }
  }
  return out;
}

template <typename T> constexpr int chunk_ch() { return Prec<T>::CC; }

template <typename T> struct PrecName;
template <> struct PrecName<float> { static const char* get() { return "f32"; } };
template <> struct PrecName<_Float16> { static const char* get() { return "f16"; } };
template <> struct PrecName<bsplit> { static const char* get() { return "bf16x3"; } };
template <> struct PrecName<fsplit> { static const char* get() { return "f16f8"; } };

// call f(T()) with the activation type of `precision`
template <typename F>
auto with_prec(int precision, F&& f) {
  if (precision == FISR_PREC_F32 || precision == FISR_PREC_F32W || precision == FISR_PREC_F32W4) return f(float());
  if (precision == FISR_PREC_F16 || precision == FISR_PREC_F16R) return f(_Float16());
  if (precision == FISR_PREC_F16F8 || precision == FISR_PREC_F16F8R) return f(fsplit());
  return f(bsplit());
}
inline bool prec_ok(int precision) {
  return precision == FISR_PREC_F32 || precision == FISR_PREC_F16 || precision == FISR_PREC_BF16X3 ||
         precision == FISR_PREC_F16F8 || precision == FISR_PREC_F32W || precision == FISR_PREC_F16R || precision == FISR_PREC_F32W4 ||
         precision == FISR_PREC_F16F8R;
}
// FISR_PREC_MIXED: which layers keep a split-precision arithmetic (f16f8) -- everything that works at the full and at
// the half resolution of level 3 (its first two encoder levels, its last two decoder levels, both heads: 55 % of the
// FLOPs); the rest of the network runs in fp16.
// (Measured with the fp64 oracle and fp16 rounding injected layer by layer: all-fp16 shifts the SR channel's PSNR by
// 0.026 dB on the default weight set, full resolution only in the split format by 0.008 dB -- 0.016 dB on the worst
// full-size window --, this plan by 0.004 dB.)
inline bool mixed_layer_is_hi(const std::string& name) {
  static const char* const hi[] = {"FISRnet/level_3/enc/level_0/", "FISRnet/level_3/enc/level_1/", "FISRnet/level_3/dec/level_1/",
                                   "FISRnet/level_3/dec/level_0/", "FISRnet/level_3/SR/",
#ifdef FISR_MIXED_FISR_SPLIT
                                   "FISRnet/level_3/FI-SR/",
#endif
  };
  for (const char* h : hi)
    if (name.compare(0, strlen(h), h) == 0) return true;
  return false;
}
inline bool prec_mixed(int precision) { return precision == FISR_PREC_MIXED || precision == FISR_PREC_MIXEDR; }
inline int layer_prec(int precision, const std::string& name) {
  if (!prec_mixed(precision)) return precision;
  if (mixed_layer_is_hi(name)) return precision == FISR_PREC_MIXED ? FISR_PREC_F16F8 : FISR_PREC_F16F8R;
  return precision == FISR_PREC_MIXED ? FISR_PREC_F16 : FISR_PREC_F16R;
}
inline bool prec_grouped16(int precision) { return precision == FISR_PREC_BF16X3 || precision == FISR_PREC_F16F8 || precision == FISR_PREC_F16F8R; }
// the precisions whose convolutions with whole 64-channel blocks run on an LDS-DMA kernel (conv3x3_dma.h / conv3x3_dma_fs.h)
inline bool prec_dma(int precision) { return precision == FISR_PREC_F16 || precision == FISR_PREC_F16F8; }

// host fp8 e4m3fn (OCP) encode, round-to-nearest-even, saturating at +-448
inline uint8_t host_fp8_e4m3(float f) {
  if (f != f) return 0x7f;
  const uint8_t sign = f < 0 ? 0x80 : 0;
  float a = std::fabs(f);
  if (a >= 448.f) return sign | 0x7e;
  if (a == 0.f) return sign;
  int ex = std::max(std::ilogb(a), -6);          // exponent of the binade (subnormals share -6)
  float q = std::ldexp(a, 3 - ex);               // [8,16) for normals, [0,8) for subnormals
  float r = std::nearbyint(q);                   // RNE in the default rounding mode
  if (r >= 16.f) { r = 8.f; ex += 1; }
  if (r < 8.f) return sign | (uint8_t)r;         // subnormal (r == 8 would have been normal)
  if (ex > 8) return sign | 0x7e;
  return sign | (uint8_t)(((ex + 7) << 3) | ((int)r - 8));
}
inline int prec_chunk(int precision) { return precision == FISR_PREC_F16 || precision == FISR_PREC_F16R ? 32 : 16; }
inline bool prec_f32w(int precision) { return precision == FISR_PREC_F32W || precision == FISR_PREC_F32W4; }   // fp32 tensors, Winograd engines
inline int prec_unit(int precision) { return precision == FISR_PREC_F32 || prec_f32w(precision) ? 4 : 8; }   // glue kernels: channels per 16 bytes
constexpr int CONV_REC = 16;   // the conv kernel stores whole 16-channel records

// host bf16 round-to-nearest-even (finite inputs)
inline uint16_t host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float host_bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Re-pack HWIO weights to [Cin/CC][9][CoutPad][64-byte record] (channel innermost) so that a
// workgroup's weight slab for one K chunk is 9 contiguous runs it copies straight to LDS.
// Record = CC values of T (float / fp16), or 16 bf16 hi followed by 16 bf16 lo (bsplit).
template <typename T>
void pack_weights(const float* w, const float* b, int ci, int co, int cin_pad, int cout_pad,
                  std::vector<char>& wp, std::vector<float>& bp, int wexp) {
  constexpr int CC = Prec<T>::CC;
  wp.assign((size_t)(cin_pad / CC) * 9 * cout_pad * CHUNK_BYTES, 0);
  bp.assign(cout_pad, 0.f);
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < ci; ++c)
      for (int n = 0; n < co; ++n) {
        const int kc = c / CC, cc = c % CC;
        // The weights are the MFMA row operand: row m of a 32-channel group ends up in accumulator
        // register r = (m&3) + 4*(m>>3) of lane half kh = (m>>2)&1.  Packing channel 16*kh + r into row m
        // makes every lane own 16 CONSECUTIVE channels (one record of the activation formats).
        // (The 16-row heads variant, cout_pad == 16, keeps the natural order: row = channel.)
        const int wi = n & 31, wk = wi >> 4, wr = wi & 15;
        const int row = cout_pad == 16 ? n : (n & ~31) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
        char* rec = wp.data() + (((size_t)kc * 9 + tap) * cout_pad + row) * CHUNK_BYTES;
        const float v = w[((size_t)tap * ci + c) * co + n];
        if constexpr (std::is_same<T, float>::value) {
          reinterpret_cast<float*>(rec)[cc] = v;
        } else if constexpr (std::is_same<T, _Float16>::value) {
          reinterpret_cast<_Float16*>(rec)[cc] = (_Float16)v;
        } else if constexpr (std::is_same<T, fsplit>::value) {
          // [ 0..31] w_h fp16 | per 8 channels {8 x fp8(w_h * 2^wexp) | 8 x fp8((w - w_h) * 2^(wexp+14))}: the layout of the
          // activations' fp8 fields (conv3x3.h), both cross terms under one block scale
          const _Float16 h = (_Float16)v;
          const float hf = (float)h;
          reinterpret_cast<_Float16*>(rec)[cc] = h;
          reinterpret_cast<uint8_t*>(rec)[32 + (cc >> 3) * 16 + (cc & 7)] = host_fp8_e4m3(std::ldexp(hf, wexp));
          reinterpret_cast<uint8_t*>(rec)[40 + (cc >> 3) * 16 + (cc & 7)] = host_fp8_e4m3(std::ldexp(v - hf, wexp + 14));
        } else {
          const uint16_t hi = host_bf16(v);
          const uint16_t lo = host_bf16(v - host_bf16_to_f32(hi));
          reinterpret_cast<uint16_t*>(rec)[cc] = hi;
          reinterpret_cast<uint16_t*>(rec)[16 + cc] = lo;
        }
      }
  for (int n = 0; n < co; ++n) bp[n] = b[n];
}

// Winograd F(2x2,3x3) weights: U = G g G^T per (ci, co), computed in double and rounded once to fp32, stored as
// the kernel's LDS image (conv3x3_wino_common.h): [Cin/8][Cout/64][position 16][row 64][32-byte record], the two 16-byte
// halves of a record swapped when bit 3 of the row is set; rows in the MFMA row order of pack_weights.
inline bool wino_eligible(int ci, int co) { (void)ci; return co >= W_BN && co % W_BN == 0; }
// the kernel addresses its input tensors, and one image of its output, with 32-bit byte offsets
inline bool wino_fits(int n, int h, int w, int c0, int c1, int co) {
  const double px = (double)n * h * w;
  return px * std::max(c0, c1) * 4.0 < 4294967296.0 && (double)h * w * co * 4.0 < 4294967296.0 - 64.0;
}
// (co need not be a multiple of the 64-channel block: the last block is zero padded, the kernel skips the stores)
void pack_weights_wino(const float* w, int ci, int co, int cin_pad, std::vector<char>& wp) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nb = (co + W_BN - 1) / W_BN, nch = cin_pad / W_CH;
  wp.assign((size_t)nch * nb * W_SLAB, 0);
  for (int c = 0; c < ci; ++c)
    for (int n = 0; n < co; ++n) {
      double g[3][3], t[4][3], u[4][4];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)(a * 3 + b) * ci + c) * co + n];
      for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) u[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      const int kc = c / W_CH, cc = c % W_CH, h = cc >> 2, e = cc & 3;
      const int blk = n / W_BN, nl = n % W_BN;
      const int wi = nl & 31, wk = wi >> 4, wr = wi & 15;
      const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
      char* slab = wp.data() + ((size_t)kc * nb + blk) * W_SLAB;
      for (int pos = 0; pos < 16; ++pos) {
        float* rec = reinterpret_cast<float*>(slab + ((size_t)pos * 64 + row) * W_REC + ((h ^ ((row >> 3) & 1)) * 16));
        rec[e] = (float)u[pos >> 2][pos & 3];
      }
    }
}

// fp16 weights for the LDS-DMA kernel (conv3x3_dma.h): [Cin/16][CoutPad/64][tap 9][row 64][32-byte record] -- a slab is the
// kernel's LDS image: rows in the MFMA row order of pack_weights, the two 16-byte halves (channels 0-7 | 8-15 of the chunk)
// swapped when bit 3 of the row is set.
void pack_weights_dma(const float* w, int ci, int co, int cin_pad, int cout_pad, std::vector<char>& wp, int bn = D_BN) {
  const int nb = cout_pad / bn, nch = cin_pad / D_CH;
  const size_t wb = (size_t)9 * bn * D_REC;
  wp.assign((size_t)nch * nb * wb, 0);
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < ci; ++c)
      for (int n = 0; n < co; ++n) {
        const int kc = c / D_CH, cc = c % D_CH, h = cc >> 3, e = cc & 7;
        const int blk = n / bn, nl = n % bn, wi = nl & 31, wk = wi >> 4, wr = wi & 15;
        const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
        char* rec = wp.data() + ((size_t)kc * nb + blk) * wb + ((size_t)tap * bn + row) * D_REC + ((h ^ ((row >> 3) & 1)) * 16);
        reinterpret_cast<_Float16*>(rec)[e] = (_Float16)w[((size_t)tap * ci + c) * co + n];
      }
}
// what the LDS-DMA kernel takes: dense or channel-range fp16 tensors whose image fits 31-bit byte offsets, both concat
// sources with the same pixel stride, whole 16-channel chunks
inline bool dma_fits(int h, int w, int c0, int c1, int cs0, int cs1) {
  return c0 % D_CH == 0 && c1 % D_CH == 0 && c0 > 0 && (c1 == 0 || cs0 == cs1) && (double)h * w * cs0 * 2.0 < 2147483648.0;
}

// f16f8 weights for the persistent LDS-DMA kernel (conv3x3_dma_fs.h): [Cin/16][Cout/64][36 864 B], a slab being the kernel's LDS
// image: [tap 9][row 64][32 B of w_h] exactly as pack_weights_dma lays fp16 out, then the fp8 parts of the tap pairs (0,3) (1,4)
// (2,5) (6,7) as [pair 4][plane 2 kh + tap of the pair][row 64][16 B] and of tap 8 as [plane kh][row 64][16 B], a 16-byte piece =
// {8 x fp8(w_h * 2^wexp) | 8 x fp8((w - w_h) * 2^(wexp+14))} of channels 8 kh .. 8 kh + 7 (pack_weights<fsplit>'s bytes 32..63).
void pack_weights_dma_fs(const float* w, int ci, int co, int cin_pad, int wexp, std::vector<char>& wp) {
  const int nb = co / FS_BN, nch = cin_pad / FS_CH;
  wp.assign((size_t)nch * nb * FS_W_BYTES, 0);
  static const int pair_of[9] = {0, 1, 2, 0, 1, 2, 3, 3, 4}, slot_of[9] = {0, 0, 0, 1, 1, 1, 0, 1, 0};
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < ci; ++c)
      for (int n = 0; n < co; ++n) {
        const int kc = c / FS_CH, cc = c % FS_CH, h = cc >> 3, e = cc & 7;
        const int blk = n / FS_BN, nl = n % FS_BN, wi = nl & 31, wk = wi >> 4, wr = wi & 15;
        const int row = (nl & 32) + (wr & 3) + 8 * (wr >> 2) + 4 * wk;
        char* slab = wp.data() + ((size_t)kc * nb + blk) * FS_W_BYTES;
        const float v = w[((size_t)tap * ci + c) * co + n];
        const _Float16 hh = (_Float16)v;
        const float hf = (float)hh;
        reinterpret_cast<_Float16*>(slab + ((size_t)tap * FS_BN + row) * 32 + ((h ^ ((row >> 3) & 1)) * 16))[e] = hh;
        const int q = pair_of[tap], t = slot_of[tap];
        uint8_t* x = reinterpret_cast<uint8_t*>(slab + FS_WM_BYTES + (q < 4 ? q * 4096 + (2 * h + t) * 1024 : 4 * 4096 + h * 1024) + row * 16);
        x[e] = host_fp8_e4m3(std::ldexp(hf, wexp));
        x[8 + e] = host_fp8_e4m3(std::ldexp(v - hf, wexp + 14));
      }
}
// what that kernel takes: dense f16f8 tensors, whole 64-channel output blocks, whole 16-channel chunks, both concat sources of the
// same width, an image inside 31-bit byte offsets
inline bool dmafs_fits(int h, int w, int c0, int c1, int co) {
  return co % FS_BN == 0 && c0 % FS_CH == 0 && c0 > 0 && (c1 == 0 || c1 == c0) && (double)h * w * c0 * 4.0 < 2147483648.0 &&
         (double)h * w * co * 4.0 < 2147483648.0;
}

// N-block of a conv: 64 channels (NT = 2), 32 (NT = 1), or the 16-row heads variant (NT = 0: Cout < 16; ConvW::rows16: fp32 also Cout = 16;
// always fp32 output; FISR_DIAG builds: FISR_CONV_HEAD16=0 turns it off for A/B runs).
template <typename T> inline int nt_for(int co) {
#ifdef FISR_DIAG
  static const bool head16 = [] { const char* e = getenv("FISR_CONV_HEAD16"); return !(e && e[0] == '0'); }();
#else
  constexpr bool head16 = true;
#endif
  if (co < 16 && head16) return 0;
  return co <= 32 ? 1 : 2;
}

template <typename T>
int upload_conv(fisr_ctx* ctx, ConvW& cw, bool wino = false, bool dma = false, bool wf4 = false) {
  constexpr int CC = Prec<T>::CC;
  cw.nt = (cw.rows16 && cw.co == 16 && std::is_same<T, float>::value) ? 0 : nt_for<T>(cw.co);
  cw.cin_pad = round_up(cw.ci, CC);
  cw.cout_pad = cw.nt == 0 ? 16 : round_up(cw.co, 32 * cw.nt);
  std::vector<char> wp;
  std::vector<float> bp;
  cw.wexp = 0;
  if (std::is_same<T, fsplit>::value) {
    float mx = 0.f;
    for (float v : cw.w) mx = std::max(mx, std::fabs(v));
    // wh8 = fp8(w_h * 2^wexp), wl8 = fp8((w - w_h) * 2^(wexp+14)): one block scale for both cross terms (conv3x3.h).  The
    // remainder is <= 2^-11 |w| for weights in fp16's normal range, so the largest power of two with max|w| * 2^(wexp+3) <= 448 (fp8
    // e4m3 max) keeps both inside the format (wh8 < 32: its 4 significant bits reach down to max|w| / 2048).  Below fp16's normal range
    // the remainder is bounded absolutely, by 2^-25: wexp <= 19 keeps 2^-25 * 2^(wexp+14) inside fp8 for layers of tiny weights too
    // (they then lose low-order bits of wh8 instead of saturating wl8).
    cw.wexp = mx > 0.f ? std::min(19, std::max(-30, 5 - std::ilogb(mx) - 1)) : 0;
  }
  pack_weights<T>(cw.w.data(), cw.b.data(), cw.ci, cw.co, cw.cin_pad, cw.cout_pad, wp, bp, cw.wexp);
  if (cw.d_w) { (void)hipFree(cw.d_w); cw.d_w = nullptr; }
  if (cw.d_b) { (void)hipFree(cw.d_b); cw.d_b = nullptr; }
  HIP_OK(ctx, hipMalloc(&cw.d_w, wp.size()));
  HIP_OK(ctx, hipMalloc((void**)&cw.d_b, bp.size() * sizeof(float)));
  HIP_OK(ctx, hipMemcpy(cw.d_w, wp.data(), wp.size(), hipMemcpyHostToDevice));
  HIP_OK(ctx, hipMemcpy(cw.d_b, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  if (cw.d_wu) { (void)hipFree(cw.d_wu); cw.d_wu = nullptr; }
  if (wino && std::is_same<T, float>::value && wino_eligible(cw.ci, cw.co)) {
    pack_weights_wino(cw.w.data(), cw.ci, cw.co, cw.cin_pad, wp);
    HIP_OK(ctx, hipMalloc(&cw.d_wu, wp.size()));
    HIP_OK(ctx, hipMemcpy(cw.d_wu, wp.data(), wp.size(), hipMemcpyHostToDevice));
  }
  if (cw.d_wu4) { (void)hipFree(cw.d_wu4); cw.d_wu4 = nullptr; }
  if (wf4 && std::is_same<T, float>::value && cw.co % F4_BN == 0 && cw.cin_pad % F4_CH == 0) {
    pack_weights_wf4(cw.w.data(), cw.ci, cw.co, cw.cin_pad, wp);
    HIP_OK(ctx, hipMalloc(&cw.d_wu4, wp.size()));
    HIP_OK(ctx, hipMemcpy(cw.d_wu4, wp.data(), wp.size(), hipMemcpyHostToDevice));
  }
  if (cw.d_wd) { (void)hipFree(cw.d_wd); cw.d_wd = nullptr; }
  if (dma && std::is_same<T, _Float16>::value && cw.co > 32) {
    cw.cout_pad_d = round_up(cw.co, D_BN);
    pack_weights_dma(cw.w.data(), cw.ci, cw.co, cw.cin_pad, cw.cout_pad_d, wp);
    HIP_OK(ctx, hipMalloc(&cw.d_wd, wp.size()));
    HIP_OK(ctx, hipMemcpy(cw.d_wd, wp.data(), wp.size(), hipMemcpyHostToDevice));
  }
  if (dma && std::is_same<T, fsplit>::value && cw.co % FS_BN == 0) {
    cw.cout_pad_d = cw.co;
    pack_weights_dma_fs(cw.w.data(), cw.ci, cw.co, cw.cin_pad, cw.wexp, wp);
    HIP_OK(ctx, hipMalloc(&cw.d_wd, wp.size()));
    HIP_OK(ctx, hipMemcpy(cw.d_wd, wp.data(), wp.size(), hipMemcpyHostToDevice));
  }
  if (cw.d_wh) { (void)hipFree(cw.d_wh); cw.d_wh = nullptr; }
  if (wino && std::is_same<T, float>::value && cw.co <= 6) {
    // [9][cin_pad][4 | 6] (outputs padded to the kernel's pairs only): 9 x 64 x 6 floats = 13.8 KB stay resident in the
    // 16 KB scalar cache the kernel reads them through; padded to 8 (18 KB) every read missed it
    const int ws = cw.co <= 4 ? 4 : 6;
    std::vector<float> wh((size_t)9 * cw.cin_pad * ws, 0.f);
    for (int tap = 0; tap < 9; ++tap)
      for (int c = 0; c < cw.ci; ++c)
        for (int n = 0; n < cw.co; ++n) wh[((size_t)tap * cw.cin_pad + c) * ws + n] = cw.w[((size_t)tap * cw.ci + c) * cw.co + n];
    HIP_OK(ctx, hipMalloc((void**)&cw.d_wh, wh.size() * sizeof(float)));
    HIP_OK(ctx, hipMemcpy(cw.d_wh, wh.data(), wh.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

// ---- kernel launch helpers ----

// Rows per wave of the conv kernel: 1 = 512-thread workgroups, 2 = 256-thread.  Measured on
// MI355X (r01): equal within 2 %; fp32 is marginally faster with 1, the 16-bit modes with 2.
// (FISR_DIAG builds: FISR_CONV_MR=1|2 in the environment forces one for A/B runs.)
template <typename T> inline int conv_mr() {
#ifdef FISR_DIAG
  static int forced = [] { const char* e = getenv("FISR_CONV_MR"); return e ? (e[0] == '2' ? 2 : 1) : 0; }();
  if (forced) return forced;
#endif
  return std::is_same<T, float>::value ? 1 : 2;
}

// ops.py:54 as a second store of the direct kernel's record epilogue: the split formats on two-row waves (conv3x3.h)
template <typename T> inline bool split_pool_ok() {
  return (std::is_same<T, fsplit>::value || std::is_same<T, bsplit>::value) && conv_mr<T>() == 2;
}
template <typename T, int NT, bool OUT_F32, int MR>
hipError_t launch_conv_variant(const ConvArgs& a, hipStream_t st) {
  static bool attr_done[64] = {};   // per device: one process may drive several GPUs (one ctx each)
  constexpr size_t lds = conv_lds_bytes<T, NT>();
  auto kern = conv3x3_mfma_kernel<T, NT, OUT_F32, MR>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  // the record stores address one output image with 32-bit offsets (buffer stores)
  if (!OUT_F32 && (double)a.H * a.W * a.Cout * sizeof(T) >= 4294967296.0 - 64.0) return hipErrorInvalidValue;
  const int tiles = ((a.W + TILE_W - 1) / TILE_W) * ((a.H + TILE_H - 1) / TILE_H) * a.N;
  dim3 grid(tiles * (a.CoutPad / (NT == 0 ? 16 : 32 * NT)));   // 1-D: the kernel orders tiles x N-blocks XCD-aware
  hipLaunchKernelGGL(kern, grid, dim3(64 * (TILE_H / MR)), lds, st, a);
  return hipGetLastError();
}

// The 3 / 6-channel heads of the fp32 engine on the vector ALU (head_conv.h); FISR_DIAG builds: FISR_HEAD_VALU=0 keeps them
// on the 16-row MFMA variant for A/B runs.
inline bool head_valu_enabled() {
#ifdef FISR_DIAG
  static const bool on = [] { const char* e = getenv("FISR_HEAD_VALU"); return !(e && e[0] == '0'); }();
  return on;
#else
  return true;
#endif
}
#ifdef FISR_DIAG
inline bool head_strip_enabled() {
  static const bool on = [] { const char* e = getenv("FISR_HEAD_STRIP"); return e && e[0] != '0'; }();
  return on;
}
#endif
hipError_t launch_head_valu(const ConvArgs& a, const float* d_wh, hipStream_t st) {
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_conv_f32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)head_lds_bytes<2>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_conv_f32_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)head_lds_bytes<3>());
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  HeadArgs h;
  h.in = (const float*)a.in0; h.w = d_wh; h.bias = a.bias; h.out = (float*)a.out;
  h.N = a.N; h.H = a.H; h.W = a.W; h.Cin = a.C0; h.Cout = a.Cout; h.relu_in = a.relu_in; h.relu_out = a.relu_out;
  h.out_cstride = a.out_cstride; h.out_coff = a.out_coff; h.out_split = a.out_split; h.out_gap = a.out_gap;
#ifdef FISR_DIAG
  if (head_strip_enabled() && head_strip_fits(a.H, a.W, a.C0)) return launch_head_strip(h, st);      // (A/B runs: diag/head_conv_strip.h)
#endif
  const int tiles = ((a.W + TILE_W - 1) / TILE_W) * ((a.H + HEAD_TH - 1) / HEAD_TH) * a.N;
  if (a.Cout <= 4) hipLaunchKernelGGL(head_conv_f32_kernel<2>, dim3(tiles), dim3(HEAD_NTHR), head_lds_bytes<2>(), st, h);
  else hipLaunchKernelGGL(head_conv_f32_kernel<3>, dim3(tiles), dim3(HEAD_NTHR), head_lds_bytes<3>(), st, h);
  return hipGetLastError();
}

// The persistent Winograd kernel (conv3x3_wino8p.h; fp32 only; a.wpk = the conv's d_wu).  Needs at least four 8-channel
// K chunks: wino_chunks_ok() is part of every caller's eligibility test.  FISR_DIAG builds: FISR_WINO_VARIANT=8 / =4 run
// the two superseded kernels instead (A/B runs).
inline bool wino_chunks_ok(int c0, int c1) { return (c0 + c1) / W_CH >= 4; }
hipError_t launch_conv_wino(const ConvArgs& a, hipStream_t st) {
  static bool attr_done[64] = {};
  static int n_cu[64] = {};
  constexpr size_t lds = wino_lds_bytes();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    const void* kerns[] = {reinterpret_cast<const void*>(conv3x3_wino8p_kernel<false, false, false>),
                           reinterpret_cast<const void*>(conv3x3_wino8p_kernel<false, false, true>),
                           reinterpret_cast<const void*>(conv3x3_wino8p_kernel<true, false, false>),
                           reinterpret_cast<const void*>(conv3x3_wino8p_kernel<true, false, true>),
                           reinterpret_cast<const void*>(conv3x3_wino8p_kernel<false, true, false>),
                           reinterpret_cast<const void*>(conv3x3_wino8p_kernel<false, true, true>),
#ifdef FISR_DIAG
                           reinterpret_cast<const void*>(conv3x3_wino_kernel),
                           reinterpret_cast<const void*>(conv3x3_wino8_kernel<false>),
                           reinterpret_cast<const void*>(conv3x3_wino8_kernel<true>),
#endif
    };
    for (const void* k : kerns) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    attr_done[dev] = true;
  }
  const int d = a.dil;
  const int tiles = (((a.W + d - 1) / d + TILE_W - 1) / TILE_W) * (((a.H + d - 1) / d + TILE_H - 1) / TILE_H) * d * d * a.N;
  const int items = tiles * (a.CoutPad / W_BN);
  // channel-range input / output, leaky relu and dilation: the GENERAL instantiation
  const bool plain = a.in0_cs == a.C0 && (a.C1 == 0 || a.in1_cs == a.C1) && a.rec_cs == a.Cout && a.rec_co == 0 &&
                     a.slope == 0.f && d == 1;
  if (d < 1 || (d > 1 && a.d2s) || !wino_chunks_ok(a.C0, a.C1)) return hipErrorInvalidValue;
  // the epilogue addresses one output (and residual) image with 32-bit byte offsets (buffer loads / stores)
  if ((double)a.H * a.W * a.rec_cs * 4.0 >= 4294967296.0 - 64.0) return hipErrorInvalidValue;
#ifdef FISR_DIAG
  static const int variant = [] { const char* e = getenv("FISR_WINO_VARIANT"); return e ? atoi(e) : 0; }();
  if (plain && variant == 4) { hipLaunchKernelGGL(conv3x3_wino_kernel, dim3(items), dim3(256), lds, st, a); return hipGetLastError(); }
  if (plain && variant == 8) {
    if (a.relu_in) hipLaunchKernelGGL(conv3x3_wino8_kernel<true>, dim3(items), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(conv3x3_wino8_kernel<false>, dim3(items), dim3(512), lds, st, a);
    return hipGetLastError();
  }
#endif
  // one workgroup per CU (the kernel needs all of a CU's LDS and half its registers), a multiple of 8 so that the
  // items of a workgroup stay on one XCD
  const int grid = std::min(items, std::max(8, n_cu[dev] & ~7));
  const bool res = a.res != nullptr;                 // (its own instantiation: see conv3x3_wino8p.h)
#define FISR_W8P_LAUNCH(RI, GEN)                                                                              \
  do {                                                                                                        \
    if (res) hipLaunchKernelGGL((conv3x3_wino8p_kernel<RI, GEN, true>), dim3(grid), dim3(512), lds, st, a, items);  \
    else hipLaunchKernelGGL((conv3x3_wino8p_kernel<RI, GEN, false>), dim3(grid), dim3(512), lds, st, a, items);     \
  } while (0)
  if (!plain) {
    if (a.relu_in) return hipErrorInvalidValue;      // (not instantiated: PWC-Net's activations come out of the producer)
    FISR_W8P_LAUNCH(false, true);
  } else if (a.relu_in) FISR_W8P_LAUNCH(true, false);
  else FISR_W8P_LAUNCH(false, false);
#undef FISR_W8P_LAUNCH
  return hipGetLastError();
}

// The LDS-DMA fp16 kernel (conv3x3_dma.h; a.wpk = the conv's d_wd, a.CoutPad = its cout_pad_d).
hipError_t launch_conv_dma(const ConvArgs& a, hipStream_t st, int nt = 2) {
  static bool attr_done[64] = {};
  constexpr size_t lds = dma_lds_bytes();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    for (const void* k : {reinterpret_cast<const void*>(conv3x3_dma_f16_kernel<false, 2>), reinterpret_cast<const void*>(conv3x3_dma_f16_kernel<true, 2>),
                          reinterpret_cast<const void*>(conv3x3_dma_f16_kernel<true, 1>)}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    attr_done[dev] = true;
  }
  const int d = a.dil;
  if (d < 1 || (d > 1 && a.d2s) || !dma_fits(a.H, a.W, a.C0, a.C1, a.in0_cs, a.in1_cs)) return hipErrorInvalidValue;
  const int tiles = (((a.W + d - 1) / d + D_TW - 1) / D_TW) * (((a.H + d - 1) / d + D_TH - 1) / D_TH) * d * d * a.N;
  const bool plain = a.in0_cs == a.C0 && (a.C1 == 0 || a.in1_cs == a.C1) && a.rec_cs == a.Cout && a.rec_co == 0 && a.slope == 0.f && d == 1;
  const dim3 grid(tiles * (a.CoutPad / (32 * nt)));
  if (nt == 1) hipLaunchKernelGGL((conv3x3_dma_f16_kernel<true, 1>), grid, dim3(256), lds, st, a);      // (the flow network's 32 / 96-channel layers)
  else if (plain) hipLaunchKernelGGL((conv3x3_dma_f16_kernel<false, 2>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((conv3x3_dma_f16_kernel<true, 2>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// The persistent f16f8 LDS-DMA kernel (conv3x3_dma_fs.h; a.wpk = the conv's d_wd, a.CoutPad = Cout).  Pooling with relu-on-load is
// not instantiated (no layer asks for it): the callers keep that combination on the direct kernel.
// The kernel reads dense tensors (pixel stride == channel count), has no dilation / leaky relu / channel-range or scattered store, and
// with depth_to_space it stores in the shuffled layout while a residual would be read in the plain one: all of that is refused here,
// so such a call falls back to the direct kernel (Runner::conv, op level) instead of computing something else.
inline bool dmafs_takes(const ConvArgs& a) {
  return !(a.pool_out && a.relu_in) && !(a.pool_out && !a.res) && a.dil == 1 && a.slope == 0.f && a.in0_cs == a.C0 && (a.C1 == 0 || a.in1_cs == a.C1) &&
         a.rec_cs == a.Cout && a.rec_co == 0 && a.out_cstride == a.Cout && a.out_coff == 0 && a.out_split >= a.Cout && !(a.d2s && a.res);
}
// Tile width: 32 everywhere since r06.  r05 kept the 8 x 64 tile for the layers with several 64-channel output blocks (half the fragment
// reads per MFMA; their half-visited lines are shared between the blocks of a tile); with two halo stages (FISR_FS_HALO2, narrow tile only:
// two wide stages do not fit beside a second workgroup) the narrow tile wins there too -- same box, micro-benchmark: 128->128 + residual
// 1305 -> 1206 us, 64->256 + d2s 4715 -> 4619, 256->128 2027 -> 1936, 64->128 730 -> 650 (128->128 relu-on-load: 1200 -> 1258, the one loss);
// `mixed` step 60.8 -> 59.9 ms, `f16f8` 84.7 -> 81.3 ms.  The wide instantiations are compiled into -DFISR_DIAG builds only
// (FISR_FS_TW=64 forces them, =1 is r05's rule, =2 wide for relu-on-load multi-block layers only: no better than 32 everywhere).
inline int dmafs_tile_w(const ConvArgs& a) {
#ifdef FISR_DIAG
  static const int forced = [] { const char* e = getenv("FISR_FS_TW"); return e ? atoi(e) : 0; }();
  if (forced == 32 || forced == 64) return forced;
  if (forced == 1) return a.Cout / FS_BN == 1 ? 32 : 64;
  if (forced == 2) return (a.relu_in && !a.res && a.Cout / FS_BN >= 2) ? 64 : 32;
#endif
  (void)a;
  return 32;
}
template <int TW>
hipError_t launch_conv_dmafs_tw(const ConvArgs& a, hipStream_t st, int n_cu, bool set_attr) {
  constexpr size_t lds = dmafs_lds_bytes(TW);
  if (set_attr) {
    for (const void* k : {reinterpret_cast<const void*>(conv3x3_dma_fs_kernel<TW, false, false, false>), reinterpret_cast<const void*>(conv3x3_dma_fs_kernel<TW, true, false, false>),
                          reinterpret_cast<const void*>(conv3x3_dma_fs_kernel<TW, false, true, false>), reinterpret_cast<const void*>(conv3x3_dma_fs_kernel<TW, true, true, false>),
                          reinterpret_cast<const void*>(conv3x3_dma_fs_kernel<TW, false, true, true>)}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
  }
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + FS_TH - 1) / FS_TH) * a.N;
  const int items = tiles * (a.Cout / FS_BN);
  // two workgroups per CU (each takes half of a CU's LDS at most), a multiple of 8 so that the items of a workgroup stay on one XCD
  const int grid = std::min(items, std::max(8, (2 * n_cu) & ~7));
  const bool res = a.res != nullptr;
  if (a.pool_out) hipLaunchKernelGGL((conv3x3_dma_fs_kernel<TW, false, true, true>), dim3(grid), dim3(256), lds, st, a, items);
  else if (a.relu_in && res) hipLaunchKernelGGL((conv3x3_dma_fs_kernel<TW, true, true, false>), dim3(grid), dim3(256), lds, st, a, items);
  else if (a.relu_in) hipLaunchKernelGGL((conv3x3_dma_fs_kernel<TW, true, false, false>), dim3(grid), dim3(256), lds, st, a, items);
  else if (res) hipLaunchKernelGGL((conv3x3_dma_fs_kernel<TW, false, true, false>), dim3(grid), dim3(256), lds, st, a, items);
  else hipLaunchKernelGGL((conv3x3_dma_fs_kernel<TW, false, false, false>), dim3(grid), dim3(256), lds, st, a, items);
  return hipGetLastError();
}
hipError_t launch_conv_dmafs(const ConvArgs& a, hipStream_t st) {
  static bool attr_done[64][2] = {};
  static int n_cu[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!n_cu[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (!dmafs_fits(a.H, a.W, a.C0, a.C1, a.Cout) || a.CoutPad != a.Cout || !dmafs_takes(a) || (a.pool_out && ((a.H | a.W) & 1))) return hipErrorInvalidValue;
  const int tw = dmafs_tile_w(a);
  bool& done = attr_done[dev][tw == 64];
#ifdef FISR_DIAG
  const hipError_t e = tw == 64 ? launch_conv_dmafs_tw<64>(a, st, n_cu[dev], !done) : launch_conv_dmafs_tw<32>(a, st, n_cu[dev], !done);
#else
  const hipError_t e = launch_conv_dmafs_tw<32>(a, st, n_cu[dev], !done);
#endif
  if (e == hipSuccess) done = true;
  return e;
}

template <typename T>
hipError_t launch_conv(const ConvArgs& a, int nt, bool out_f32, hipStream_t st) {
  const bool m1 = conv_mr<T>() == 1;
  if (nt == 0) {
    return m1 ? launch_conv_variant<T, 0, true, 1>(a, st) : launch_conv_variant<T, 0, true, 2>(a, st);
  }
  if (nt == 1) {
    if (out_f32) return m1 ? launch_conv_variant<T, 1, true, 1>(a, st) : launch_conv_variant<T, 1, true, 2>(a, st);
    return m1 ? launch_conv_variant<T, 1, false, 1>(a, st) : launch_conv_variant<T, 1, false, 2>(a, st);
  }
  if (out_f32) return m1 ? launch_conv_variant<T, 2, true, 1>(a, st) : launch_conv_variant<T, 2, true, 2>(a, st);
  return m1 ? launch_conv_variant<T, 2, false, 1>(a, st) : launch_conv_variant<T, 2, false, 2>(a, st);
}

int prof_class(fisr_ctx* ctx, const std::string& name) {
  for (size_t i = 0; i < ctx->prof_names.size(); ++i)
    if (ctx->prof_names[i] == name) return (int)i;
  ctx->prof_names.push_back(name);
  ctx->prof_ms.push_back(0);
  ctx->prof_flops.push_back(0);
  ctx->prof_bytes.push_back(0);
  ctx->prof_launches.push_back(0);
  return (int)ctx->prof_names.size() - 1;
}

hipEvent_t prof_event(fisr_ctx* ctx) {
  if (ctx->ev_used == ctx->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    ctx->ev_pool.push_back(e);
  }
  return ctx->ev_pool[ctx->ev_used++];
}

struct ProfScope {
  fisr_ctx* ctx;
  hipStream_t st;
  bool on;
  ProfEntry pe;
  ProfScope(fisr_ctx* c, hipStream_t s, const std::string& cls, double flops, double bytes)
      : ctx(c), st(s), on(c && c->prof) {
    if (!on) return;
    pe.cls = prof_class(ctx, cls);
    pe.flops = flops;
    pe.bytes = bytes;
    pe.a = prof_event(ctx);
    pe.b = prof_event(ctx);
    (void)hipEventRecord(pe.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(pe.b, st);
    ctx->prof_entries.push_back(pe);
  }
};

int prof_flush(fisr_ctx* ctx) {
  for (auto& pe : ctx->prof_entries) {
    (void)hipEventSynchronize(pe.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, pe.a, pe.b);
    ctx->prof_ms[pe.cls] += ms;
    ctx->prof_flops[pe.cls] += pe.flops;
    ctx->prof_bytes[pe.cls] += pe.bytes;
    ctx->prof_launches[pe.cls] += 1;
  }
  ctx->prof_entries.clear();
  ctx->ev_used = 0;
  return 0;
}

ColorConsts make_color_consts() {
  // utils.py:106-110 / warp script :35-40 (YUV2RGB) and :48-52 (RGB2YUV)
  const double tinv[3][3] = {{0.00456621, 0., 0.00625893},
                             {0.00456621, -0.00153632, -0.00318811},
                             {0.00456621, 0.00791071, 0.}};
  const double tf[3][3] = {{65.481, 128.553, 24.966}, {-37.797, -74.203, 112}, {112, -93.786, -18.214}};
  ColorConsts cc;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      cc.t[i][j] = 255 * tinv[i][j];
      cc.f[i][j] = tf[i][j] / 255;
    }
    // offset = (255*Tinv) @ [16,128,128]^T
    volatile double s = cc.t[i][0] * 16.0;
    s = s + cc.t[i][1] * 128.0;
    s = s + cc.t[i][2] * 128.0;
    cc.off[i] = s;
  }
  return cc;
}

// ---- the forward schedule ----

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = false;
  void* alloc(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    peak = std::max(peak, off);
    return dry ? nullptr : base + a;
  }
};

template <typename T>
struct Runner {
  fisr_ctx* ctx;
  hipStream_t st;
  Arena ar;
  int rc = 0;

  T* talloc(size_t elems) { return (T*)ar.alloc(elems * sizeof(T)); }

  void check(hipError_t e, const char* what) {
    if (e != hipSuccess && rc == 0) rc = fail(ctx, FISR_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  }

  // ops.py:7-11 (+ fused neighbours, see conv3x3.h)
  // (pool_out: the 2x2 max pooling of the output as a second store of the conv -- only when pool_fuses() says this conv runs on the
  //  F(4x4) Winograd kernel)
  bool pool_fuses(const std::string& name, int c, int h, int w) {
    auto it = ctx->convs.find(name);
    // (r04: the split formats' direct kernel stores the pooled map too -- conv3x3.h, MR = 2 record store)
    if (split_pool_ok<T>()) return it != ctx->convs.end() && it->second.nt >= 1 && !(h & 1) && !(w & 1);
    return std::is_same<T, float>::value && ctx->wf4 && it != ctx->convs.end() && it->second.d_wu4 && wf4_fits(h, w, c, 0, it->second.co) &&
           wf4_wins(h, w, c) && !(h & 1) && !(w & 1);
  }
  // (ups: in0 is the half-resolution map and the x2 bilinear of ops.py:69 happens on the conv's way into LDS -- only when up_fuses()
  //  says this conv runs on the F(4x4) Winograd kernel; h, w are the ENLARGED map's)
  bool up_fuses(const std::string& name, int c, int h, int w) { return std::is_same<T, float>::value && pool_fuses(name, c, h, w); }      // (the same conditions)
  void conv(const std::string& name, const T* in0, int c0, const T* in1, int c1, const T* res, void* out,
            int n, int h, int w, int flags, bool out_f32 = false, int cstride = 0, int coff = 0,
            int split = 1 << 30, int gap = 0, void* pool_out = nullptr, bool ups = false) {
    if (rc) return;
    auto it = ctx->convs.find(name);
    if (it == ctx->convs.end()) { rc = fail(ctx, FISR_EMISSING, "unknown conv " + name); return; }
    const ConvW& cw = it->second;
    if (with_prec(cw.prec, [](auto tag) { return !std::is_same<decltype(tag), T>::value; })) {
      rc = fail(ctx, FISR_ESTATE, name + ": weights are packed for another precision than this engine stage runs in");
      return;
    }
    if (c0 + c1 != cw.cin_pad) {
      rc = fail(ctx, FISR_EINVAL, name + ": channel mismatch " + std::to_string(c0 + c1) + " vs " + std::to_string(cw.cin_pad));
      return;
    }
    if (ar.dry) return;
    ConvArgs a;
    a.in0 = in0; a.in1 = in1; a.wpk = cw.d_w; a.bias = cw.d_b; a.res = res; a.out = out;
    a.C0 = c0; a.C1 = c1; a.N = n; a.H = h; a.W = w; a.Cout = cw.co; a.CoutPad = cw.cout_pad;
    a.in0_cs = c0; a.in1_cs = c1; a.rec_cs = cw.co; a.rec_co = 0; a.slope = 0.f; a.dil = 1;
    a.relu_in = (flags & FISR_CONV_RELU_IN) != 0;
    a.relu_out = (flags & FISR_CONV_RELU_OUT) != 0;
    a.d2s = (flags & FISR_CONV_D2S) != 0;
    a.d2s_shift = a.d2s ? ilog2(cw.co / 4) : 0;
    a.out_cstride = cstride ? cstride : cw.co;
    a.out_coff = coff; a.out_split = split; a.out_gap = gap; a.trace = nullptr; a.wexp = cw.wexp;
    const double px = (double)n * h * w;
    const bool use_wino = std::is_same<T, float>::value && ctx->wino && cw.d_wu && !out_f32 && wino_chunks_ok(c0, c1) && wino_fits(n, h, w, c0, c1, cw.co);
    if (use_wino) a.wpk = cw.d_wu;
    const bool use_wf4 = std::is_same<T, float>::value && ctx->wf4 && cw.d_wu4 && !out_f32 && wf4_fits(h, w, c0, c1, cw.co) && wf4_wins(h, w, c0 + c1);
    if (use_wf4) a.wpk = cw.d_wu4;
    if (pool_out) {
      if (!use_wf4 && !(split_pool_ok<T>() && cw.nt >= 1 && !out_f32 && !a.d2s && !(h & 1) && !(w & 1))) {
        rc = fail(ctx, FISR_ESTATE, name + ": fused pooling asked of a conv that runs on neither the F(4x4) kernel nor the split formats' record store");
        return;
      }
      a.pool_out = pool_out;
    }
    if (ups) {
      if (!use_wf4) { rc = fail(ctx, FISR_ESTATE, name + ": fused up-sampling asked of a conv that does not run on the F(4x4) kernel"); return; }
      a.ups = 1;
    }
    const bool use_head = std::is_same<T, float>::value && ctx->wino && out_f32 && cw.d_wh && c1 == 0 && c0 % HEAD_CH == 0 && !res && head_valu_enabled();
    const bool use_dma = std::is_same<T, _Float16>::value && cw.d_wd && !out_f32 && dma_fits(h, w, c0, c1, c0, c1);
    const bool use_dmafs = std::is_same<T, fsplit>::value && cw.d_wd && !out_f32 && dmafs_fits(h, w, c0, c1, cw.co) && dmafs_takes(a);
    if (use_dma || use_dmafs) { a.wpk = cw.d_wd; a.CoutPad = cw.cout_pad_d; }
    char cls[96];
    if (use_dma) snprintf(cls, sizeof cls, "conv3x3_dma<f16>");
    else if (use_dmafs) snprintf(cls, sizeof cls, "conv3x3_dma_fs<f16f8,tw%d,%s,%s>", dmafs_tile_w(a), a.relu_in ? "relu_in" : "plain", pool_out ? "res+pool" : res ? "res" : "nores");
    else if (use_wf4) snprintf(cls, sizeof cls, "conv3x3_wf4<f32w4,%s,%s>", a.relu_in ? "relu_in" : ups ? "up2" : "plain", pool_out ? "res+pool" : res ? "res" : "nores");
    else if (use_wino) snprintf(cls, sizeof cls, "conv3x3_wino8p<f32w,%s,%s>", a.relu_in ? "relu_in" : "plain", res ? "res" : "nores");
    else if (use_head) snprintf(cls, sizeof cls, "head_conv_f32<valu>");
    else snprintf(cls, sizeof cls, "conv3x3_mfma<%s,NT%d>%s", PrecName<T>::get(), cw.nt, out_f32 ? "_f32out" : "");
    std::string cname(cls);
    if (ctx->prof_mode == 2) {
      char shp[64];
      snprintf(shp, sizeof shp, " %dx%dx%d %d->%d", n, h, w, c0 + c1, cw.co);
      cname = name.substr(name.find("level_")) + shp;
    }
    ProfScope ps(ctx, st, cname, 2.0 * 9 * cw.ci * cw.co * px,
                 px * (double)((ups ? c0 * 0.25 : c0) + c1 + cw.co + (res ? cw.co : 0)) * sizeof(T));
    check(use_dma ? launch_conv_dma(a, st)
                  : use_dmafs ? launch_conv_dmafs(a, st)
                  : use_wf4 ? launch_conv_wf4(a, st)
                  : use_wino ? launch_conv_wino(a, st) : (use_head ? launch_head_valu(a, cw.d_wh, st) : launch_conv<T>(a, cw.nt, out_f32, st)), name.c_str());
  }

  // ops.py:39-44 res_block, in place on X with scratch A.
  void rb(const std::string& name, T* X, T* A, int c, int n, int h, int w, bool relu_out, void* pool_out = nullptr) {
    conv(name + "/conv/0", X, c, nullptr, 0, nullptr, A, n, h, w, FISR_CONV_RELU_IN | FISR_CONV_RELU_OUT);
    conv(name + "/conv/1", A, c, nullptr, 0, X, X, n, h, w, relu_out ? FISR_CONV_RELU_OUT : 0, false, 0, 0, 1 << 30, 0, pool_out);
  }

  void pool(const T* in, T* out, int n, int h, int w, int c) {  // ops.py:54
    if (rc || ar.dry) return;
    const size_t work = (size_t)n * (h / 2) * (w / 2) * c / Unit<T>::UC;
    ProfScope ps(ctx, st, "maxpool2", 0, (double)n * h * w * c * sizeof(T) * 1.25);
    hipLaunchKernelGGL(maxpool2_kernel<T>, dim3(grid_for(work)), dim3(256), 0, st, in, out, n, h, w, c);
    check(hipGetLastError(), "maxpool2");
  }
  void up(const T* in, T* out, int n, int h, int w, int c) {  // ops.py:69
    if (rc || ar.dry) return;
    const size_t work = (size_t)n * ((h + UP_ROWS - 1) / UP_ROWS) * w * c / Unit<T>::UC;   // one thread per unit of a 16-row column segment
    ProfScope ps(ctx, st, "upsample2", 0, (double)n * h * w * c * sizeof(T) * 5.0);
    hipLaunchKernelGGL(upsample2_kernel<T>, dim3(grid_for(work)), dim3(256), 0, st, in, out, n, h, w, c);
    check(hipGetLastError(), "upsample2");
  }
  // fisr_forward_frames: the level inputs come straight from the windows' source planes (prep_level_frames_kernel); img is unused
  const FrameItems* fsrc = nullptr;
  void prep(const float* img, const float* pred, T* out, int n, int H, int W, int s, int cpad) {
    if (rc || ar.dry) return;
    const size_t work = (size_t)n * (H / s) * (W / s) * (cpad / 16);   // one thread per 16-channel record
    if (fsrc) {
      // bytes: 9 u8 + 4 x 8 + 4 x 12 source bytes per output pixel, the prediction, the records
      const double opx = (double)n * (H / s) * (W / s);
      ProfScope ps(ctx, st, "prep_level_frames", 0, opx * (89.0 + (pred ? 36.0 : 0.0) + (double)cpad * sizeof(T)));
      const int spans = (W / s + 255) / 256;
      hipLaunchKernelGGL(prep_level_frames_kernel<T>, dim3(std::min(n * (H / s) * spans, 1 << 20)), dim3(256), 0, st, *fsrc, pred, out, n, H, W, s, cpad);
      check(hipGetLastError(), "prep_level_frames");
      return;
    }
    ProfScope ps(ctx, st, s == 1 ? "prep_level_input_s1" : "prep_level_input", 0, (double)work * 16 * (4 + sizeof(T)));
    if (s == 1)      // (level 3: the staged variant; the strided levels read too sparsely for it)
      hipLaunchKernelGGL(prep_level_input_s1_kernel<T>, dim3(grid_for((size_t)n * H * W)), dim3(256), 0, st, img, pred, out, (size_t)n * H * W, cpad);
    else
      if (s == 2 || s == 4) {
        const int spans = (W / s + 256 / s - 1) / (256 / s);
        hipLaunchKernelGGL(prep_level_input_rows_kernel<T>, dim3(std::min(n * (H / s) * spans, 1 << 20)), dim3(256), 0, st, img, pred, out, n, H, W, s, cpad);
      } else
        hipLaunchKernelGGL(prep_level_input_kernel<T>, dim3(grid_for(work)), dim3(256), 0, st, img, pred, out, n, H, W, s, cpad);
    check(hipGetLastError(), "prep_level_input");
  }

  // The pieces of one level (FISRnet.py:83-108); `level` strings them together, the mixed-precision engine runs them
  // on two Runners (see MixedRunner).
  static const int* widths() { static const int ch[3] = {64, 128, 256}; return ch; }

  // Enc_level_res ops.py:48-55 at h x w: returns the pooled map (h/2 x w/2), *skip = relu(res_block(...))
  T* enc_level(const std::string& P, int l, const T* cur, int cc, int n, int h, int w, T** skip) {
    const int c = widths()[l];
    const std::string e = P + "/enc/level_" + std::to_string(l);
    const size_t px = (size_t)n * h * w;
    T* X = talloc(px * c);
    T* A = talloc(px * c);
    conv(e + "/conv/0", cur, cc, nullptr, 0, nullptr, X, n, h, w, 0);
    rb(e + "/res_block/0", X, A, c, n, h, w, false);
    T* Pl = talloc(px / 4 * c);
    // ops.py:54: the pooled map is a second store of the level's last conv where that conv runs on the F(4x4) kernel (r03)
    const bool fused = pool_fuses(e + "/res_block/1/conv/1", c, h, w);
    rb(e + "/res_block/1", X, A, c, n, h, w, true, fused ? Pl : nullptr);  // n = relu(res_block(...)); skip = n
    *skip = X;
    if (!fused) pool(X, Pl, n, h, w, c);
    return Pl;
  }
  // Bottleneck_res ops.py:59-63
  T* bottleneck(const std::string& P, const T* cur, int cc, int n, int h, int w) {
    const size_t px = (size_t)n * h * w;
    T* X = talloc(px * 512);
    T* A = talloc(px * 512);
    conv(P + "/bottleneck/conv/0", cur, cc, nullptr, 0, nullptr, X, n, h, w, 0);
    rb(P + "/bottleneck/res_block/0", X, A, 512, n, h, w, true);
    return X;
  }
  // Dec_level_res ops.py:67-76: cur is h x w with cc channels, the result 2h x 2w with widths()[l]
  T* dec_level(const std::string& P, int l, const T* cur, int cc, const T* skip, int n, int h, int w) {
    const int c = widths()[l];
    const std::string d = P + "/dec/level_" + std::to_string(l);
    // tf.image.resize_images + Conv2d: on the F(4x4) kernel the enlarged map never exists (conv3x3_wf4.h, UPS)
    const bool fused = up_fuses(d + "/resize", cc, 2 * h, 2 * w);
    T* U = fused ? nullptr : talloc((size_t)n * h * w * 4 * cc);
    if (!fused) up(cur, U, n, h, w, cc);
    h *= 2; w *= 2;
    const size_t px = (size_t)n * h * w;
    T* D = talloc(px * c);
    conv(d + "/resize", fused ? cur : U, cc, nullptr, 0, nullptr, D, n, h, w, FISR_CONV_RELU_OUT, false, 0, 0, 1 << 30, 0, nullptr, fused);
    T* X = talloc(px * c);
    T* A = talloc(px * c);
    conv(d + "/conv/0", D, c, skip, c, nullptr, X, n, h, w, 0);  // concat([n, skip])
    rb(d + "/res_block/0", X, A, c, n, h, w, false);
    rb(d + "/res_block/1", X, A, c, n, h, w, true);
    return X;
  }
  // one head FISRnet.py:95-100 (hd = 0: FI-SR) / :101-106 (hd = 1: SR): cur [n,h,w,64] -> its channels of pred float32 [n,2h,2w,9];
  // Hx, A: [n,h,w,64] scratch, S: [n,2h,2w,64] scratch
  void head(const std::string& P, int hd, const T* cur, int n, int h, int w, float* pred, T* Hx, T* A, T* S) {
    const std::string p = P + (hd == 0 ? "/FI-SR" : "/SR");
    conv(p + "/conv/0", cur, 64, nullptr, 0, nullptr, Hx, n, h, w, 0);
    rb(p + "/res_block/0", Hx, A, 64, n, h, w, false);
    conv(p + "/conv/1", Hx, 64, nullptr, 0, nullptr, S, n, h, w,
         FISR_CONV_RELU_IN | FISR_CONV_RELU_OUT | FISR_CONV_D2S);
    // pred = concat([fr1, SR, fr2]): FI-SR channels 0-2 -> 0-2, 3-5 -> 6-8; SR -> 3-5
    if (hd == 0) conv(p + "/conv/2", S, 64, nullptr, 0, nullptr, pred, n, 2 * h, 2 * w, 0, true, 9, 0, 3, 3);
    else         conv(p + "/conv/2", S, 64, nullptr, 0, nullptr, pred, n, 2 * h, 2 * w, 0, true, 9, 3);
  }
  // heads FISRnet.py:95-108: cur [n,h,w,64] -> pred float32 [n,2h,2w,9]
  void heads(const std::string& P, const T* cur, int n, int h, int w, float* pred) {
    const size_t px = (size_t)n * h * w;
    T* Hx = talloc(px * 64);
    T* A = talloc(px * 64);
    T* S = talloc(px * 4 * 64);
    for (int hd = 0; hd < 2; ++hd) head(P, hd, cur, n, h, w, pred, Hx, A, S);
  }

  // One U-Net + the two heads at working resolution rh x rw (FISRnet.py:83-108).
  // xin: [n,rh,rw,cin_pad]; pred: float32 [n,2rh,2rw,9].
  void level(int lv, const T* xin, int cin_pad, int n, int rh, int rw, float* pred) {
    const std::string P = "FISRnet/level_" + std::to_string(lv);
    const T* cur = xin;
    int cc = cin_pad, h = rh, w = rw;
    T* skip[3];
    for (int l = 0; l < 3; ++l) {
      cur = enc_level(P, l, cur, cc, n, h, w, &skip[l]);
      cc = widths()[l]; h /= 2; w /= 2;
    }
    cur = bottleneck(P, cur, cc, n, h, w);
    cc = 512;
    for (int l = 2; l >= 0; --l) {
      cur = dec_level(P, l, cur, cc, skip[l], n, h, w);
      cc = widths()[l]; h *= 2; w *= 2;
    }
    heads(P, cur, n, h, w, pred);
  }

  // FISRnet.py:73-173
  int forward(const float* in, int n, int h, int w, float* l3, float* l2, float* l1) {
    constexpr int CC = Prec<T>::CC;
    if (!l1) l1 = (float*)ar.alloc((size_t)n * (h / 2) * (w / 2) * 9 * sizeof(float));
    if (!l2) l2 = (float*)ar.alloc((size_t)n * h * w * 9 * sizeof(float));
    const size_t mark = ar.off;
    const int c1 = round_up(29, CC), c23 = round_up(38, CC);
    {
      T* x = talloc((size_t)n * (h / 4) * (w / 4) * c1);
      prep(in, nullptr, x, n, h, w, 4, c1);
      level(1, x, c1, n, h / 4, w / 4, l1);
    }
    ar.off = mark;
    {
      T* x = talloc((size_t)n * (h / 2) * (w / 2) * c23);
      prep(in, l1, x, n, h, w, 2, c23);
      level(2, x, c23, n, h / 2, w / 2, l2);
    }
    ar.off = mark;
    {
      T* x = talloc((size_t)n * h * w * c23);
      prep(in, l2, x, n, h, w, 1, c23);
      level(3, x, c23, n, h, w, l3);
    }
    return rc;
  }
};

// FISR_PREC_MIXED: fp16 everywhere except at the full resolution of level 3 (mixed_layer_is_hi), which stays in
// a split format (f16f8: 17 % faster than split bf16 at the same accuracy for this purpose).  Two Runners over ONE arena; the activation format changes twice, both times on a quarter-size tensor:
// behind the pooling of level 3's first encoder level (f16f8 -> fp16) and in front of the x2 up-sampling of its
// last decoder level (fp16 -> f16f8).
struct MixedRunner {
  typedef fsplit THi;             // the split format of the full-resolution stage (FISR_PREC_F16F8: the fastest of the two)
  Runner<_Float16> lo;
  Runner<THi> hi;

  template <typename TI, typename TO>
  void convert(Runner<TO>& dst, const TI* in, TO* out, size_t elems) {
    if (dst.rc || dst.ar.dry) return;
    const size_t nrec = elems / 16;
    ProfScope ps(dst.ctx, dst.st, "convert_records", 0, (double)elems * (sizeof(TI) + sizeof(TO)));
    hipLaunchKernelGGL((convert_records_kernel<TI, TO>), dim3(grid_for(nrec)), dim3(256), 0, dst.st, in, out, nrec);
    dst.check(hipGetLastError(), "convert_records");
  }

  int forward(const float* in, int n, int h, int w, float* l3, float* l2, float* l1) {
    if (!l1) l1 = (float*)lo.ar.alloc((size_t)n * (h / 2) * (w / 2) * 9 * sizeof(float));
    if (!l2) l2 = (float*)lo.ar.alloc((size_t)n * h * w * 9 * sizeof(float));
    const size_t mark = lo.ar.off;
    const int c1 = round_up(29, Prec<_Float16>::CC), c2 = round_up(38, Prec<_Float16>::CC), c3 = round_up(38, Prec<THi>::CC);
    {
      _Float16* x = lo.talloc((size_t)n * (h / 4) * (w / 4) * c1);
      lo.prep(in, nullptr, x, n, h, w, 4, c1);
      lo.level(1, x, c1, n, h / 4, w / 4, l1);
    }
    lo.ar.off = mark;
    {
      _Float16* x = lo.talloc((size_t)n * (h / 2) * (w / 2) * c2);
      lo.prep(in, l1, x, n, h, w, 2, c2);
      lo.level(2, x, c2, n, h / 2, w / 2, l2);
    }
    lo.ar.off = mark;
    if (lo.rc) return lo.rc;
    const std::string P = "FISRnet/level_3";
    hi.ar = lo.ar;
    THi* x = hi.talloc((size_t)n * h * w * c3);
    hi.prep(in, l2, x, n, h, w, 1, c3);
    THi* skiph[2] = {nullptr, nullptr};
    const THi* pooled = hi.enc_level(P, 0, x, c3, n, h, w, &skiph[0]);
    pooled = hi.enc_level(P, 1, pooled, 64, n, h / 2, w / 2, &skiph[1]);
    lo.ar = hi.ar;
    const size_t px4 = (size_t)n * (h / 4) * (w / 4);
    _Float16* cur16 = lo.talloc(px4 * 128);
    convert(lo, pooled, cur16, px4 * 128);
    _Float16* skip2 = nullptr;
    const _Float16* cur = lo.enc_level(P, 2, cur16, 128, n, h / 4, w / 4, &skip2);
    cur = lo.bottleneck(P, cur, 256, n, h / 8, w / 8);
    cur = lo.dec_level(P, 2, cur, 512, skip2, n, h / 8, w / 8);          // -> 256 channels at h/4 x w/4
    if (lo.rc) return lo.rc;
    hi.ar = lo.ar;
    THi* curb = hi.talloc(px4 * 256);
    convert(hi, cur, curb, px4 * 256);
    const THi* top = hi.dec_level(P, 1, curb, 256, skiph[1], n, h / 4, w / 4);   // -> 128 channels at h/2 x w/2
    top = hi.dec_level(P, 0, top, 128, skiph[0], n, h / 2, w / 2);
    // r04: the FI-SR branch (FISRnet.py:157-162) in fp16 -- its frames sit at the 37.86 dB operating point (README.md:97), where the
    // branch's fp16 rounding shifts the PSNR by a tenth of what it would on the 48.07 dB SR frame, and nothing of it reaches the SR
    // branch; the decoder output is converted once (2 B per channel for the branch's five convs instead of 4)
    const size_t px = (size_t)n * h * w;
    THi* Hx = hi.talloc(px * 64);
    THi* A = hi.talloc(px * 64);
    THi* S = hi.talloc(px * 4 * 64);
    hi.head(P, 1, top, n, h, w, l3, Hx, A, S);
    if (hi.rc) return hi.rc;
    lo.ar = hi.ar;
    // (the fp16 branch works inside the SR branch's scratch: S is dead once SR/conv/2 has been enqueued on the same stream)
#ifdef FISR_MIXED_FISR_SPLIT
    // A/B build (lib.build(defines=["FISR_MIXED_FISR_SPLIT"])): the FI-SR branch stays in the split format, as before r04 -- kept until
    // --check_published has been run on the real checkpoint (the fp16 branch's accuracy was measured with synthetic weights)
    hi.head(P, 0, top, n, h, w, l3, Hx, A, S);
    lo.ar = hi.ar;
    return hi.rc;
#endif
    static_assert(sizeof(THi) * 256 >= sizeof(_Float16) * 448, "the fp16 FI-SR branch lives inside the SR branch's scratch S: [4 px, 64] THi must hold [px, 448] fp16");
    _Float16* s16 = (_Float16*)S;                  // px * 4 * 64 fsplit = 8 x px x 64 fp16
    _Float16* top16 = s16;                         // [px, 64]
    _Float16* Hx16 = s16 + px * 64;
    _Float16* A16 = s16 + px * 128;
    _Float16* S16 = s16 + px * 192;                // [4 px, 64] -> ends at px * 448 <= px * 512
    convert(lo, top, top16, px * 64);
    lo.head(P, 0, top16, n, h, w, l3, Hx16, A16, S16);
    return lo.rc;
  }
};

inline size_t ws_bytes_mixed(fisr_ctx* ctx, int n, int h, int w) {
  MixedRunner r;
  r.lo.ctx = r.hi.ctx = ctx; r.lo.st = r.hi.st = nullptr; r.lo.ar.dry = true;
  r.forward(nullptr, n, h, w, (float*)1, nullptr, nullptr);
  return r.lo.ar.peak + 256;
}

template <typename T>
size_t ws_bytes_t(fisr_ctx* ctx, int n, int h, int w) {
  Runner<T> r;
  r.ctx = ctx; r.st = nullptr; r.ar.dry = true;
  r.forward(nullptr, n, h, w, (float*)1, nullptr, nullptr);
  return r.ar.peak + 256;
}

}  // namespace

// =============================== C ABI ===============================

extern "C" {

// FISR_SRC_HASH: sha256 (first 16 hex digits) over csrc/ + include/fisr.h, passed by the build (fisr_amd/lib.py) so that a
// bench line or a profile can be tied to the sources of the binary that produced it
#ifndef FISR_SRC_HASH
#define FISR_SRC_HASH "unhashed"
#endif
#ifdef FISR_DIAG
const char* fisr_version(void) { return "fisr_hip 0.3 (gfx950) src " FISR_SRC_HASH " DIAG"; }
#else
const char* fisr_version(void) { return "fisr_hip 0.3 (gfx950) src " FISR_SRC_HASH; }
#endif

const char* fisr_last_error(const fisr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int fisr_create(fisr_ctx** out, int device_id) {
  if (!out) return fail(nullptr, FISR_EINVAL, "fisr_create: out is NULL");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, FISR_EHIP, std::string("fisr_create: no HIP device (") + hipGetErrorString(e) + ")");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, FISR_EINVAL, "fisr_create: bad device id");
  DeviceGuard guard(device_id);
  HIP_OK(nullptr, guard.err);
  fisr_ctx* c = new fisr_ctx();
  c->dev = device_id;
  for (auto& s : all_specs()) {
    ConvW cw;
    cw.ci = s.ci; cw.co = s.co;
    c->convs[s.name] = cw;
  }
  if (hipMalloc((void**)&c->d_scalar, 64) != hipSuccess) { delete c; return fail(nullptr, FISR_EHIP, "hipMalloc"); }
  *out = c;
  return 0;
}

void fisr_destroy(fisr_ctx* ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->dev);
  for (auto& kv : ctx->convs) {
    if (kv.second.d_w) (void)hipFree(kv.second.d_w);
    if (kv.second.d_b) (void)hipFree(kv.second.d_b);
    if (kv.second.d_wu) (void)hipFree(kv.second.d_wu);
    if (kv.second.d_wu4) (void)hipFree(kv.second.d_wu4);
    if (kv.second.d_wh) (void)hipFree(kv.second.d_wh);
    if (kv.second.d_wd) (void)hipFree(kv.second.d_wd);
  }
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->d_scalar) (void)hipFree(ctx->d_scalar);
  delete ctx;
}

int fisr_set_weight(fisr_ctx* ctx, const char* name, const float* host, const int64_t* shape, int rank) {
  if (!ctx || !name || !host || !shape) return fail(ctx, FISR_EINVAL, "fisr_set_weight: null argument");
  std::string s(name);
  // TF names may carry a ":0" suffix
  const size_t colon = s.rfind(':');
  if (colon != std::string::npos) s = s.substr(0, colon);
  if (s.size() < 3) return 1;
  const std::string kind = s.substr(s.size() - 2);
  const std::string base = s.substr(0, s.size() - 2);
  auto it = ctx->convs.find(base);
  if (it == ctx->convs.end() || (kind != "/w" && kind != "/b")) return 1;  // not ours (Adam slot, step, ...)
  ConvW& cw = it->second;
  if (kind == "/w") {
    if (rank != 4 || shape[0] != 3 || shape[1] != 3 || shape[2] != cw.ci || shape[3] != cw.co)
      return fail(ctx, FISR_EINVAL, s + ": expected shape [3,3," + std::to_string(cw.ci) + "," + std::to_string(cw.co) + "]");
    cw.w.assign(host, host + (size_t)9 * cw.ci * cw.co);
    cw.have_w = true;
  } else {
    if (rank != 1 || shape[0] != cw.co)
      return fail(ctx, FISR_EINVAL, s + ": expected shape [" + std::to_string(cw.co) + "]");
    cw.b.assign(host, host + cw.co);
    cw.have_b = true;
  }
  ctx->finalized = false;
  return 0;
}

int fisr_num_variables_set(const fisr_ctx* ctx) {
  if (!ctx) return 0;
  int k = 0;
  for (auto& kv : ctx->convs) k += (int)kv.second.have_w + (int)kv.second.have_b;
  return k;
}

int fisr_finalize_weights(fisr_ctx* ctx, int precision) {
  if (!ctx) return fail(nullptr, FISR_EINVAL, "fisr_finalize_weights: ctx is NULL");
  if (!prec_ok(precision) && precision != FISR_PREC_MIXED && precision != FISR_PREC_MIXEDR) return fail(ctx, FISR_EINVAL, "fisr_finalize_weights: unknown precision");
#ifndef FISR_DIAG
  // The A/B engines (superseded kernels everywhere) exist in diagnostics builds only; their ids stay valid at op level, where
  // they name a kernel the product still runs on the layers its successors do not take.
  if (precision == FISR_PREC_F32W || precision == FISR_PREC_F16R || precision == FISR_PREC_F16F8R || precision == FISR_PREC_MIXEDR)
    return fail(ctx, FISR_EINVAL, "fisr_finalize_weights: FISR_PREC_F32W / F16R / F16F8R / MIXEDR are A/B engines of the diagnostics build (-DFISR_DIAG); "
                                  "the shipped engines are FISR_PREC_F32W4, F32, BF16X3, F16F8, F16, MIXED");
#endif
  for (auto& s : all_specs()) {
    const ConvW& cw = ctx->convs[s.name];
    if (!cw.have_w) return fail(ctx, FISR_EMISSING, "missing variable " + s.name + "/w");
    if (!cw.have_b) return fail(ctx, FISR_EMISSING, "missing variable " + s.name + "/b");
  }
  DeviceGuard guard(ctx->dev);
  HIP_OK(ctx, guard.err);
  for (auto& kv : ctx->convs) {
    const int lp = layer_prec(precision, kv.first);
    int rc = with_prec(lp, [&](auto tag) { return upload_conv<decltype(tag)>(ctx, kv.second, prec_f32w(lp), prec_dma(lp), lp == FISR_PREC_F32W4); });
    if (rc) return rc;
    kv.second.prec = lp;
  }
  ctx->wino = prec_f32w(precision);
  ctx->wf4 = precision == FISR_PREC_F32W4;
  ctx->precision = precision;
  ctx->finalized = true;
  return 0;
}

size_t fisr_workspace_bytes(const fisr_ctx* cctx, int n, int h, int w) {
  fisr_ctx* ctx = const_cast<fisr_ctx*>(cctx);
  if (!ctx || !ctx->finalized || n < 1 || h < 32 || w < 32 || h % 32 || w % 32) return 0;
  if (prec_mixed(ctx->precision)) return ws_bytes_mixed(ctx, n, h, w);
  return with_prec(ctx->precision, [&](auto tag) { return ws_bytes_t<decltype(tag)>(ctx, n, h, w); });
}

int fisr_forward(fisr_ctx* ctx, const float* in, int n, int h, int w, float* out_l3, float* out_l2,
                 float* out_l1, void* workspace, size_t workspace_bytes, void* stream) {
  if (!ctx) return fail(nullptr, FISR_EINVAL, "fisr_forward: ctx is NULL");
  if (!ctx->finalized) return fail(ctx, FISR_ESTATE, "fisr_forward: weights not finalized");
  if (!in || !out_l3 || !workspace) return fail(ctx, FISR_EINVAL, "fisr_forward: null tensor");
  if (n < 1 || h < 32 || w < 32 || h % 32 || w % 32)
    return fail(ctx, FISR_EINVAL, "fisr_forward: h and w must be positive multiples of 32 (FISRnet.py:820-824)");
  const size_t need = fisr_workspace_bytes(ctx, n, h, w);
  if (workspace_bytes < need)
    return fail(ctx, FISR_ENOMEM, "fisr_forward: workspace " + std::to_string(workspace_bytes) + " < " + std::to_string(need));
  DeviceGuard guard(ctx->dev);
  HIP_OK(ctx, guard.err);
  hipStream_t st = (hipStream_t)stream;
  auto run = [&](auto tag) -> int {
    typedef decltype(tag) T;
    Runner<T> r;
    r.ctx = ctx; r.st = st;
    r.ar.base = (char*)workspace; r.ar.cap = workspace_bytes;
    return r.forward(in, n, h, w, out_l3, out_l2, out_l1);
  };
  if (prec_mixed(ctx->precision)) {
    MixedRunner r;
    r.lo.ctx = r.hi.ctx = ctx; r.lo.st = r.hi.st = st;
    r.lo.ar.base = (char*)workspace; r.lo.ar.cap = workspace_bytes;
    return r.forward(in, n, h, w, out_l3, out_l2, out_l1);
  }
  return with_prec(ctx->precision, run);
}

int fisr_forward_frames(fisr_ctx* ctx, const fisr_src_item* items, int n, int h0, int w0, int h, int w, float* out_l3, float* out_l2,
                        float* out_l1, void* workspace, size_t workspace_bytes, void* stream) {
  if (!ctx) return fail(nullptr, FISR_EINVAL, "fisr_forward_frames: ctx is NULL");
  if (!ctx->finalized) return fail(ctx, FISR_ESTATE, "fisr_forward_frames: weights not finalized");
  if (!items || !out_l3 || !workspace) return fail(ctx, FISR_EINVAL, "fisr_forward_frames: null argument");
  if (n < 1 || n > FISR_MAX_SRC_ITEMS)
    return fail(ctx, FISR_EINVAL, "fisr_forward_frames: 1 <= n <= FISR_MAX_SRC_ITEMS (" + std::to_string(FISR_MAX_SRC_ITEMS) + ") items per call");
  if (h < 32 || w < 32 || h % 32 || w % 32)
    return fail(ctx, FISR_EINVAL, "fisr_forward_frames: h and w must be positive multiples of 32 (FISRnet.py:820-824)");
  static_assert(SRC_MAX_ITEMS == FISR_MAX_SRC_ITEMS, "the kernel-argument item table and the header's limit");
  FrameItems fi;
  memset(&fi, 0, sizeof(fi));
  fi.W0 = w0;
  for (int i = 0; i < n; ++i) {
    const fisr_src_item& it = items[i];
    if (it.y0 < 0 || it.x0 < 0 || it.y0 + h > h0 || it.x0 + w > w0)
      return fail(ctx, FISR_EINVAL, "fisr_forward_frames: item " + std::to_string(i) + " does not lie inside the " + std::to_string(h0) + " x " +
                                        std::to_string(w0) + " frame");
    for (int k = 0; k < 3; ++k) {
      if (!it.frames[k]) return fail(ctx, FISR_EINVAL, "fisr_forward_frames: null frame pointer");
      fi.pp[i].fr[k] = it.frames[k];
    }
    for (int k = 0; k < 4; ++k) {
      if (!it.flows[k] || !it.warps[k] || ((size_t)it.flows[k] & 7) || ((size_t)it.warps[k] & 3))
        return fail(ctx, FISR_EINVAL, "fisr_forward_frames: flow planes must be non-null and 8-byte aligned, warp planes 4-byte aligned");
      fi.pp[i].fl[k] = it.flows[k];
      fi.pp[i].wp[k] = it.warps[k];
    }
    fi.y0[i] = it.y0; fi.x0[i] = it.x0;
  }
  const size_t need = fisr_workspace_bytes(ctx, n, h, w);
  if (workspace_bytes < need)
    return fail(ctx, FISR_ENOMEM, "fisr_forward_frames: workspace " + std::to_string(workspace_bytes) + " < " + std::to_string(need));
  DeviceGuard guard(ctx->dev);
  HIP_OK(ctx, guard.err);
  hipStream_t st = (hipStream_t)stream;
  auto run = [&](auto tag) -> int {
    typedef decltype(tag) T;
    Runner<T> r;
    r.ctx = ctx; r.st = st; r.fsrc = &fi;
    r.ar.base = (char*)workspace; r.ar.cap = workspace_bytes;
    return r.forward(nullptr, n, h, w, out_l3, out_l2, out_l1);
  };
  if (prec_mixed(ctx->precision)) {
    MixedRunner r;
    r.lo.ctx = r.hi.ctx = ctx; r.lo.st = r.hi.st = st; r.lo.fsrc = r.hi.fsrc = &fi;
    r.lo.ar.base = (char*)workspace; r.lo.ar.cap = workspace_bytes;
    return r.forward(nullptr, n, h, w, out_l3, out_l2, out_l1);
  }
  return with_prec(ctx->precision, run);
}

int fisr_profile_enable(fisr_ctx* ctx, int on) {
  if (!ctx) return FISR_EINVAL;
  ctx->prof = on != 0;
  ctx->prof_mode = on;
  return 0;
}
int fisr_profile_reset(fisr_ctx* ctx) {
  if (!ctx) return FISR_EINVAL;
  prof_flush(ctx);
  for (size_t i = 0; i < ctx->prof_names.size(); ++i) {
    ctx->prof_ms[i] = 0; ctx->prof_flops[i] = 0; ctx->prof_bytes[i] = 0; ctx->prof_launches[i] = 0;
  }
  return 0;
}
int fisr_profile_read(fisr_ctx* ctx, int cap, const char** name, double* total_ms, int64_t* launches,
                      double* flops, double* bytes) {
  if (!ctx) return FISR_EINVAL;
  prof_flush(ctx);
  const int k = (int)ctx->prof_names.size();
  for (int i = 0; i < k && i < cap; ++i) {
    if (name) name[i] = ctx->prof_names[i].c_str();
    if (total_ms) total_ms[i] = ctx->prof_ms[i];
    if (launches) launches[i] = ctx->prof_launches[i];
    if (flops) flops[i] = ctx->prof_flops[i];
    if (bytes) bytes[i] = ctx->prof_bytes[i];
  }
  return k;
}

// ---- glue ----

int fisr_warp(const float* src, const float* flow, float scale, int h, int w, int quantized, float* dst, void* stream) {
  if (!src || !flow || !dst || h < 1 || w < 1) return fail(nullptr, FISR_EINVAL, "fisr_warp: bad argument");
  static const ColorConsts cc = make_color_consts();
  DeviceGuard guard(device_of(dst));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(warp_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, (hipStream_t)stream, src, flow, scale,
                     h, w, quantized, dst, cc);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

int fisr_pack_input(const uint8_t* const* fr, const float* const* fl, const float* const* wp, int h0, int w0,
                    int h, int w, float* out, void* stream) {
  if (!fr || !fl || !wp || !out || h > h0 || w > w0 || h < 1 || w < 1)
    return fail(nullptr, FISR_EINVAL, "fisr_pack_input: bad argument");
  PackPtrs pp;
  for (int i = 0; i < 3; ++i) pp.fr[i] = fr[i];
  for (int i = 0; i < 4; ++i) { pp.fl[i] = fl[i]; pp.wp[i] = wp[i]; }
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, (hipStream_t)stream, pp, h0, w0, h, w, out);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

int fisr_unpack_output(const float* pred, int h, int w, uint8_t* yuv_u8, uint8_t* rgb_u8, void* stream) {
  if (!pred || h < 1 || w < 1) return fail(nullptr, FISR_EINVAL, "fisr_unpack_output: bad argument");
  static const ColorConsts cc = make_color_consts();
  DeviceGuard guard(device_of(pred));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(unpack_output_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, (hipStream_t)stream, pred, h, w,
                     yuv_u8, rgb_u8, cc);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

int fisr_stitch(const float* tile, int th, int tw, int sy, int sx, int ch, int cw, float* full, int fh, int fw,
                int dy, int dx, void* stream) {
  if (!tile || !full || sy < 0 || sx < 0 || sy + ch > th || sx + cw > tw || dy < 0 || dx < 0 || dy + ch > fh || dx + cw > fw)
    return fail(nullptr, FISR_EINVAL, "fisr_stitch: region out of range");
  DeviceGuard guard(device_of(full));
  HIP_OK(nullptr, guard.err);
  hipLaunchKernelGGL(stitch_kernel, dim3(grid_for((size_t)ch * cw * 9)), dim3(256), 0, (hipStream_t)stream, tile, tw, sy,
                     sx, ch, cw, full, fw, dy, dx);
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

// One 8-byte accumulator per device for the two reductions below, allocated on first use and kept (a
// hipMalloc/hipFree pair per call is an implicit device-wide synchronisation).  The calls end with a stream
// synchronise, so one slot per device is enough for the single-threaded hosts this ABI serves.
static double* reduce_scratch(int dev) {
  static double* slot[64] = {};
  if (dev < 0 || dev >= 64) return nullptr;
  if (!slot[dev] && hipMalloc((void**)&slot[dev], sizeof(double)) != hipSuccess) slot[dev] = nullptr;
  return slot[dev];
}

int fisr_sse_vs_u8(const float* pred, const uint8_t* gt, size_t count, double* out_host, void* stream) {
  if (!pred || !gt || !out_host) return fail(nullptr, FISR_EINVAL, "fisr_sse_vs_u8: null argument");
  const int dev = device_of(pred);
  DeviceGuard guard(dev);
  HIP_OK(nullptr, guard.err);
  double* d = reduce_scratch(dev);
  if (!d) return fail(nullptr, FISR_EHIP, "fisr_sse_vs_u8: pred is not a device pointer / no scratch");
  hipStream_t st = (hipStream_t)stream;
  HIP_OK(nullptr, hipMemsetAsync(d, 0, sizeof(double), st));
  hipLaunchKernelGGL(sse_u8_kernel, dim3(grid_for(count)), dim3(256), 0, st, pred, gt, count, d);
  HIP_OK(nullptr, hipGetLastError());
  HIP_OK(nullptr, hipMemcpyAsync(out_host, d, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(nullptr, hipStreamSynchronize(st));
  return 0;
}

int fisr_ssim_u8(const uint8_t* a, const uint8_t* b, int h, int w, int cstride, int coff, double* out_host, void* stream) {
  if (!a || !b || !out_host || h < 7 || w < 7 || cstride < 3 || coff < 0 || coff + 3 > cstride)
    return fail(nullptr, FISR_EINVAL, "fisr_ssim_u8: bad argument");
  const int dev = device_of(a);
  DeviceGuard guard(dev);
  HIP_OK(nullptr, guard.err);
  double* d = reduce_scratch(dev);
  if (!d) return fail(nullptr, FISR_EHIP, "fisr_ssim_u8: a is not a device pointer / no scratch");
  hipStream_t st = (hipStream_t)stream;
  HIP_OK(nullptr, hipMemsetAsync(d, 0, sizeof(double), st));
  const size_t tiles = (size_t)(h / 7) * (w / 7) * 3;
  hipLaunchKernelGGL(ssim_tiles_kernel, dim3(grid_for(tiles)), dim3(256), 0, st, a, b, h, w, cstride, coff, 7, d);
  HIP_OK(nullptr, hipGetLastError());
  double sum = 0;
  HIP_OK(nullptr, hipMemcpyAsync(&sum, d, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_OK(nullptr, hipStreamSynchronize(st));
  *out_host = sum / (double)tiles;
  return 0;
}

// ---- op-level entries ----

}  // extern "C"
// (pool_out: the second store of conv3x3_wf4.h's POOL instantiation -- fisr_op_conv3x3_pool)
static int op_conv3x3_impl(const void* in0, int c0, const void* in1, int c1, const float* w_host, const float* b_host,
                           int cout, const void* res, void* out, void* pool_out, int n, int h, int w, int flags, int precision,
                           int out_f32, void* stream) {
  if (!in0 || !w_host || !b_host || !out || n < 1 || h < 1 || w < 1 || cout < 1)
    return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: bad argument");
  // (r04: the split formats' direct kernel has the pooled second store too -- any flags but d2s / fused bilinear, residual optional)
  const bool pool_split = precision == FISR_PREC_BF16X3 || precision == FISR_PREC_F16F8 || precision == FISR_PREC_F16F8R;
  if (pool_out && pool_split && (out_f32 || (h & 1) || (w & 1) || (flags & (FISR_CONV_D2S | FISR_CONV_UP2_IN)) || cout % CONV_REC))
    return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3_pool: on the split formats the pooled second store needs even h / w, whole 16-channel records, no d2s / fused bilinear");
  if (pool_out && !pool_split && (precision != FISR_PREC_F32W4 || out_f32 || !res || (h & 1) || (w & 1) || (flags & (FISR_CONV_RELU_IN | FISR_CONV_D2S | FISR_CONV_UP2_IN)) ||
                   !wf4_fits(h, w, c0, c1, cout)))
    return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3_pool: the pooled second store exists on the F(4x4) kernel only (FISR_PREC_F32W4, even h / w, "
                                      "a residual input, no relu-on-load / d2s / fused bilinear; ops.py:52-54)");
  if (!prec_ok(precision)) return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: unknown precision");
  const int cc = prec_chunk(precision);
  if (cout % CONV_REC) {
    // partial 16-channel records only exist on the fp32-output heads (channel-scatter store, no residual)
    if (precision == FISR_PREC_F32 || prec_f32w(precision)) out_f32 = 1;
    if (!out_f32) return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: cout must be a multiple of 16 unless out_f32");
  }
  if (out_f32 && (res || (flags & FISR_CONV_D2S)))
    return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: the fp32-output store has no residual / d2s");
  // (FISR_PREC_F16: the LDS-DMA kernel, which takes the convolutions with Cout > 32, works in 16-channel chunks)
  const bool dma_op = precision == FISR_PREC_F16 && cout > 32 && !out_f32 && dma_fits(h, w, c0, c1, c0, c1);
  if (((c0 % cc || c1 % cc) && !dma_op) || (c1 && !in1)) return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: channels must be multiples of the chunk");
  if ((flags & FISR_CONV_D2S) && (cout % 4 || !is_pow2(cout / 4) || cout / 4 < CONV_REC))
    return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: d2s needs cout/4 to be a power of two >= 16 (whole 16-channel records)");
  if ((flags & FISR_CONV_UP2_IN) && (precision != FISR_PREC_F32W4 || out_f32 || !wf4_fits(h, w, c0, c1, cout) || (h & 1) || (w & 1) || c1 || res ||
                                     (flags & (FISR_CONV_RELU_IN | FISR_CONV_D2S))))
    return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3: FISR_CONV_UP2_IN is fused on the F(4x4) kernel only (FISR_PREC_F32W4, even h / w, one source, no residual, relu-on-load or d2s)");
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  ConvW cw;
  cw.ci = c0 + c1; cw.co = cout;
  cw.w.assign(w_host, w_host + (size_t)9 * cw.ci * cout);
  cw.b.assign(b_host, b_host + cout);
  int rc = with_prec(precision, [&](auto tag) { return upload_conv<decltype(tag)>(nullptr, cw, prec_f32w(precision), prec_dma(precision), precision == FISR_PREC_F32W4); });
  if (rc) return rc;
  // (FISR_PREC_F32W4 at op level: the F(4x4) kernel for every shape it takes -- the engine adds its map-size rule, wf4_wins)
  const bool use_wf4 = precision == FISR_PREC_F32W4 && cw.d_wu4 && !out_f32 && wf4_fits(h, w, c0, c1, cout) && !(res && (flags & FISR_CONV_D2S));
  const bool use_wino = !use_wf4 && prec_f32w(precision) && cw.d_wu && !out_f32 && wino_chunks_ok(c0, c1) && wino_fits(n, h, w, c0, c1, cout);
  const bool use_dma = precision == FISR_PREC_F16 && cw.d_wd && !out_f32 && dma_fits(h, w, c0, c1, c0, c1);
  // (FISR_PREC_F16F8: the persistent LDS-DMA kernel wherever it has an instantiation)
  bool use_dmafs = precision == FISR_PREC_F16F8 && cw.d_wd && !out_f32 && dmafs_fits(h, w, c0, c1, cout) &&
                   !(pool_out && ((flags & FISR_CONV_RELU_IN) || !res)) && !((flags & FISR_CONV_D2S) && res);
  ConvArgs a;
  a.in0 = in0; a.in1 = in1; a.wpk = use_dma || use_dmafs ? cw.d_wd : use_wf4 ? cw.d_wu4 : (use_wino ? cw.d_wu : cw.d_w); a.bias = cw.d_b; a.res = res; a.out = out;
  a.C0 = c0; a.C1 = c1; a.N = n; a.H = h; a.W = w; a.Cout = cout; a.CoutPad = use_dma || use_dmafs ? cw.cout_pad_d : cw.cout_pad;
  a.in0_cs = c0; a.in1_cs = c1; a.rec_cs = cout; a.rec_co = 0; a.slope = 0.f; a.dil = 1;
  a.relu_in = (flags & FISR_CONV_RELU_IN) != 0;
  a.relu_out = (flags & FISR_CONV_RELU_OUT) != 0;
  a.d2s = (flags & FISR_CONV_D2S) != 0;
  a.d2s_shift = a.d2s ? ilog2(cout / 4) : 0;
  a.ups = (flags & FISR_CONV_UP2_IN) != 0;
  a.pool_out = pool_out;
  a.out_cstride = cout; a.out_coff = 0; a.out_split = 1 << 30; a.out_gap = 0; a.trace = nullptr; a.wexp = cw.wexp;
  hipStream_t st = (hipStream_t)stream;
  // (the fp32 engine's 3 / 6-channel heads: the vector-ALU kernel, as in the forward)
  const bool use_head = prec_f32w(precision) && out_f32 && cw.d_wh && c1 == 0 && c0 % HEAD_CH == 0 && !res && head_valu_enabled();
  hipError_t e = use_dma ? launch_conv_dma(a, st)
                 : use_dmafs ? launch_conv_dmafs(a, st)
                 : use_wf4 ? launch_conv_wf4(a, st)
                 : use_wino ? launch_conv_wino(a, st)
                 : use_head ? launch_head_valu(a, cw.d_wh, st)
                            : with_prec(precision, [&](auto tag) { return launch_conv<decltype(tag)>(a, cw.nt, out_f32 != 0, st); });
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(cw.d_w);
  (void)hipFree(cw.d_b);
  if (cw.d_wu) (void)hipFree(cw.d_wu);
  if (cw.d_wu4) (void)hipFree(cw.d_wu4);
  if (cw.d_wh) (void)hipFree(cw.d_wh);
  if (cw.d_wd) (void)hipFree(cw.d_wd);
  if (e != hipSuccess) return fail(nullptr, FISR_EHIP, std::string("conv launch: ") + hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(nullptr, FISR_EHIP, std::string("conv sync: ") + hipGetErrorString(e2));
  return 0;
}
extern "C" {

int fisr_op_conv3x3(const void* in0, int c0, const void* in1, int c1, const float* w_host, const float* b_host,
                    int cout, const void* res, void* out, int n, int h, int w, int flags, int precision,
                    int out_f32, void* stream) {
  return op_conv3x3_impl(in0, c0, in1, c1, w_host, b_host, cout, res, out, nullptr, n, h, w, flags, precision, out_f32, stream);
}

int fisr_op_conv3x3_pool(const void* in0, int c0, const void* in1, int c1, const float* w_host, const float* b_host,
                         int cout, const void* res, void* out, void* pool_out, int n, int h, int w, int flags, int precision, void* stream) {
  if (!pool_out) return fail(nullptr, FISR_EINVAL, "fisr_op_conv3x3_pool: pool_out is NULL");
  return op_conv3x3_impl(in0, c0, in1, c1, w_host, b_host, cout, res, out, pool_out, n, h, w, flags, precision, 0, stream);
}

int fisr_op_maxpool2(const void* in, void* out, int n, int h, int w, int c, int precision, void* stream) {
  if (!in || !out || h % 2 || w % 2 || !prec_ok(precision)) return fail(nullptr, FISR_EINVAL, "fisr_op_maxpool2: bad argument");
  const int uc = prec_unit(precision);
  if (c % (prec_grouped16(precision) ? 16 : uc)) return fail(nullptr, FISR_EINVAL, "fisr_op_maxpool2: c must be a multiple of the channel unit");
  const size_t work = (size_t)n * (h / 2) * (w / 2) * c / uc;
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  with_prec(precision, [&](auto tag) {
    typedef decltype(tag) T;
    hipLaunchKernelGGL(maxpool2_kernel<T>, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, n, h, w, c);
    return 0;
  });
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

// The input of a level's first convolution (FISRnet.py:81, 112-113, 144): img [n,h,w,29] sub-sampled by s (1 | 2 | 4: legacy BICUBIC at an
// integer factor) ++ pred [n,h/s,w/s,9] (nullable) ++ zeros up to cpad channels, as float32 records -- the kernels Runner::prep launches
int fisr_op_prep_level_input(const float* img, const float* pred, float* out, int n, int h, int w, int s, int cpad, void* stream) {
  if (!img || !out || n < 1 || h < 1 || w < 1 || (s != 1 && s != 2 && s != 4) || h % s || w % s || cpad % 16 || cpad < (pred ? 38 : 29))
    return fail(nullptr, FISR_EINVAL, "fisr_op_prep_level_input: bad argument");
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  Runner<float> r;
  r.ctx = nullptr; r.st = (hipStream_t)stream;
  r.prep(img, pred, out, n, h, w, s, cpad);
  HIP_OK(nullptr, hipGetLastError());
  return r.rc;
}

int fisr_op_upsample2(const void* in, void* out, int n, int h, int w, int c, int precision, void* stream) {
  if (!in || !out || !prec_ok(precision)) return fail(nullptr, FISR_EINVAL, "fisr_op_upsample2: bad argument");
  const int uc = prec_unit(precision);
  if (c % (prec_grouped16(precision) ? 16 : uc)) return fail(nullptr, FISR_EINVAL, "fisr_op_upsample2: c must be a multiple of the channel unit");
  const size_t work = (size_t)n * ((h + UP_ROWS - 1) / UP_ROWS) * w * c / uc;
  DeviceGuard guard(device_of(out));
  HIP_OK(nullptr, guard.err);
  with_prec(precision, [&](auto tag) {
    typedef decltype(tag) T;
    hipLaunchKernelGGL(upsample2_kernel<T>, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, (const T*)in, (T*)out, n, h, w, c);
    return 0;
  });
  HIP_OK(nullptr, hipGetLastError());
  return 0;
}

// Micro-benchmark of one conv shape (not part of the product path): allocates its own buffers,
// runs `iters` launches back to back on the default stream and returns the mean microseconds per
// launch (HIP events) in *out_us.  Weights/activations are pseudo-random bit patterns.
static int bench_conv_impl(int precision, int n, int h, int w, int cin, int cout, int flags, int with_res, int iters,
                           double* out_us, bool zero_fill, unsigned lomask, const char* trace_file) {
  if (!prec_ok(precision) || !out_us || iters < 1) return fail(nullptr, FISR_EINVAL, "fisr_bench_conv: bad argument");
  const int cc = prec_chunk(precision);
  if (cin % cc || cout % 8) return fail(nullptr, FISR_EINVAL, "fisr_bench_conv: channels must be whole chunks");
  const size_t abytes = precision == FISR_PREC_F16 || precision == FISR_PREC_F16R ? 2 : 4;
  const size_t in_b = (size_t)n * h * w * cin * abytes, out_b = (size_t)n * h * w * cout * abytes;
  ConvW cw;
  cw.ci = cin; cw.co = cout;
  cw.w.resize((size_t)9 * cin * cout);
  cw.b.assign(cout, 0.01f);
  uint32_t st = 12345u;
  for (auto& v : cw.w) { st = st * 1664525u + 1013904223u; v = zero_fill ? 0.f : ((int)(st >> 9) % 2001 - 1000) * 2e-5f; }
  int rc = with_prec(precision, [&](auto tag) { return upload_conv<decltype(tag)>(nullptr, cw, prec_f32w(precision), prec_dma(precision), precision == FISR_PREC_F32W4); });
  if (rc) return rc;
  const bool use_wf4 = precision == FISR_PREC_F32W4 && cw.d_wu4 && wf4_fits(h, w, cin, 0, cout) && !(with_res && (flags & FISR_CONV_D2S));
  const bool use_wino = !use_wf4 && prec_f32w(precision) && cw.d_wu && wino_chunks_ok(cin, 0) && wino_fits(n, h, w, cin, 0, cout);
  const bool use_dma = precision == FISR_PREC_F16 && cw.d_wd && dma_fits(h, w, cin, 0, cin, 0);
  const bool use_dmafs = precision == FISR_PREC_F16F8 && cw.d_wd && dmafs_fits(h, w, cin, 0, cout);
  void *d_in = nullptr, *d_out = nullptr, *d_res = nullptr;
  HIP_OK(nullptr, hipMalloc(&d_in, in_b));
  HIP_OK(nullptr, hipMalloc(&d_out, out_b));
  if (with_res) HIP_OK(nullptr, hipMalloc(&d_res, out_b));
  {
    std::vector<uint16_t> hbuf(1 << 20);
    for (auto& v : hbuf) { st = st * 1664525u + 1013904223u; v = zero_fill ? 0 : ((uint16_t)(0x3c00 + ((st >> 12) & 0x3ff)) ^ (uint16_t)((st >> 31) << 15)); }
    if (lomask != 0xffffu)   // DVFS probe: fewer toggling bits in the lo planes
      for (size_t i = 0; i < hbuf.size(); ++i) if ((i >> 4) & 1) hbuf[i] &= (uint16_t)lomask;
    for (size_t o = 0; o < in_b; o += hbuf.size() * 2)
      HIP_OK(nullptr, hipMemcpy((char*)d_in + o, hbuf.data(), std::min(hbuf.size() * 2, in_b - o), hipMemcpyHostToDevice));
    if (d_res)
      for (size_t o = 0; o < out_b; o += hbuf.size() * 2)
        HIP_OK(nullptr, hipMemcpy((char*)d_res + o, hbuf.data(), std::min(hbuf.size() * 2, out_b - o), hipMemcpyHostToDevice));
  }
  ConvArgs a;
  a.in0 = d_in; a.in1 = nullptr; a.wpk = use_dma || use_dmafs ? cw.d_wd : use_wf4 ? cw.d_wu4 : (use_wino ? cw.d_wu : cw.d_w); a.bias = cw.d_b; a.res = d_res; a.out = d_out;
  a.C0 = cin; a.C1 = 0; a.N = n; a.H = h; a.W = w; a.Cout = cout; a.CoutPad = use_dma || use_dmafs ? cw.cout_pad_d : cw.cout_pad;
  a.in0_cs = cin; a.in1_cs = 0; a.rec_cs = cout; a.rec_co = 0; a.slope = 0.f; a.dil = 1;
  a.relu_in = (flags & FISR_CONV_RELU_IN) != 0;
  a.relu_out = (flags & FISR_CONV_RELU_OUT) != 0;
  a.d2s = (flags & FISR_CONV_D2S) != 0;
  a.d2s_shift = a.d2s ? ilog2(cout / 4) : 0;
  a.out_cstride = cout; a.out_coff = 0; a.out_split = 1 << 30; a.out_gap = 0; a.trace = nullptr; a.wexp = cw.wexp;
  unsigned long long* d_trace = nullptr;
  const size_t nblocks = use_dmafs ? (size_t)1024      // (persistent: one trace record per workgroup)
                         : use_dma ? (size_t)(((w + D_TW - 1) / D_TW) * ((h + D_TH - 1) / D_TH) * n) * (cw.cout_pad_d / D_BN)
                         : use_wf4 ? (size_t)(((w + F4_TW - 1) / F4_TW) * ((h + F4_TH - 1) / F4_TH) * n) * (cw.cout_pad / F4_BN)
                                 : (size_t)(((w + TILE_W - 1) / TILE_W) * ((h + TILE_H - 1) / TILE_H) * n) *
                                       (use_wino ? cw.cout_pad / W_BN : cw.cout_pad / (cw.nt ? 32 * cw.nt : 16));
  // (conv3x3_dma_fs.h writes FS_TRACE_WORDS words per workgroup -- its grid is at most 2 workgroups per CU --, the other kernels 8)
  const size_t trace_rec = use_dmafs ? (size_t)FS_TRACE_WORDS * 8 : 64;
  if (trace_file) {
    HIP_OK(nullptr, hipMalloc((void**)&d_trace, nblocks * trace_rec));
    HIP_OK(nullptr, hipMemset(d_trace, 0, nblocks * trace_rec));
    a.trace = d_trace;
  }
  hipEvent_t e0, e1;
  HIP_OK(nullptr, hipEventCreate(&e0));
  HIP_OK(nullptr, hipEventCreate(&e1));
  hipError_t e = hipSuccess;
  auto launch = [&]() -> hipError_t {
    if (use_dma) return launch_conv_dma(a, nullptr);
    if (use_dmafs) return launch_conv_dmafs(a, nullptr);
    if (use_wf4) return launch_conv_wf4(a, nullptr);
    if (use_wino) return launch_conv_wino(a, nullptr);
    return with_prec(precision, [&](auto tag) { return launch_conv<decltype(tag)>(a, cw.nt, false, nullptr); });
  };
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch();
  HIP_OK(nullptr, hipDeviceSynchronize());
  HIP_OK(nullptr, hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch();
  HIP_OK(nullptr, hipEventRecord(e1, nullptr));
  HIP_OK(nullptr, hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(nullptr, hipEventElapsedTime(&ms, e0, e1));
  *out_us = (double)ms * 1e3 / iters;
  if (trace_file) {
    std::vector<unsigned long long> tr(nblocks * trace_rec / 8);
    HIP_OK(nullptr, hipMemcpy(tr.data(), d_trace, nblocks * trace_rec, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_file, "wb")) { fwrite(tr.data(), 8, tr.size(), f); fclose(f); }
    (void)hipFree(d_trace);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(d_in); (void)hipFree(d_out); if (d_res) (void)hipFree(d_res);
  (void)hipFree(cw.d_w); (void)hipFree(cw.d_b); if (cw.d_wu) (void)hipFree(cw.d_wu); if (cw.d_wu4) (void)hipFree(cw.d_wu4); if (cw.d_wh) (void)hipFree(cw.d_wh); if (cw.d_wd) (void)hipFree(cw.d_wd);
  if (e != hipSuccess) return fail(nullptr, FISR_EHIP, std::string("bench launch: ") + hipGetErrorString(e));
  return 0;
}

int fisr_bench_conv(int precision, int n, int h, int w, int cin, int cout, int flags, int with_res, int iters, double* out_us) {
  return bench_conv_impl(precision, n, h, w, cin, cout, flags, with_res, iters, out_us, false, 0xffffu, nullptr);
}

#ifdef FISR_DIAG
// FISR_DIAG builds only: the micro-benchmark with zero-filled operands / masked lo planes (DVFS probes: zero operands draw
// less power) and a per-workgroup {start, main-loop end, end, HW_ID, ...} timeline written to trace_file (scripts/trace_conv.py)
FISR_API int fisr_diag_bench_conv(int precision, int n, int h, int w, int cin, int cout, int flags, int with_res, int iters, double* out_us,
                         int zero_fill, unsigned lomask, const char* trace_file) {
  return bench_conv_impl(precision, n, h, w, cin, cout, flags, with_res, iters, out_us, zero_fill != 0, lomask, trace_file);
}
#endif

}  // extern "C"

#include "fisr_comm.h"
#include "fisr_pwc.h"
#include "fisr_train.h"
