// Probe (diagnostics): operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 A and B).
// D[i][j] = sum_k A[i][k]*B[k][j] * 2^(sa-127) * 2^(sb-127) per 32-wide K block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const unsigned char* A /*[64 lanes][32 B]*/, const unsigned char* B, const int* sa, const int* sb,
                      float* D /*[64 lanes][16]*/) {
  const int l = threadIdx.x;
  i32x8 a, b;
  memcpy(&a, A + l * 32, 32);
  memcpy(&b, B + l * 32, 32);
  f32x16 c;
  for (int k = 0; k < 16; ++k) c[k] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int k = 0; k < 16; ++k) D[l * 16 + k] = c[k];
}

static unsigned char fp8(float v) {  // exact small values only: 0, 1, 2, 0.5
  if (v == 0) return 0;
  if (v == 1) return 0x38;
  if (v == 2) return 0x40;
  if (v == 0.5f) return 0x30;
  if (v == 4) return 0x48;
  return 0x38;
}

int main() {
  unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 4096);
  std::vector<unsigned char> A(2048), B(2048);
  std::vector<int> sa(64), sb(64);
  std::vector<float> D(1024);
  auto run = [&](const char* what) {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    printf("== %s\n", what);
  };
  auto outrc = [&](int lane, int reg, int& row, int& col) { col = lane & 31; row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); };
  // Test 1: A only lane la byte ba = 1.0 ; B all ones; scales 127.  Expect row (la&31) of D all = 1 if that (lane,byte) is a valid k.
  for (int t = 0; t < 4; ++t) {
    int la = (t & 1) ? 37 : 5, ba = (t & 2) ? 19 : 0;
    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0x38); std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
    A[la * 32 + ba] = 0x38;
    run("T1: one A element, B ones");
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) if (D[lane * 16 + r] != 0) { int row, col; outrc(lane, r, row, col); if (col == 0) printf("  A(lane %d, byte %d) -> D row %d = %g\n", la, ba, row, D[lane * 16 + r]); }
  }
  // Test 2: which B (lane, byte) pairs with A (lane la, byte ba): A one element; B one element = 2.0 at (lb, bb) scanning
  {
    int la = 5, ba = 3;
    for (int lb_half = 0; lb_half < 2; ++lb_half) {
      int hits = 0;
      for (int bb = 0; bb < 32; ++bb) {
        std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0); A[la * 32 + ba] = 0x38;
        int lb = 9 + 32 * lb_half; B[lb * 32 + bb] = 0x40;
        run("");
        for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) if (D[lane * 16 + r] != 0) { int row, col; outrc(lane, r, row, col); printf("  A(l%d,b%d) x B(l%d,b%d) -> D[%d][%d] = %g\n", la, ba, lb, bb, row, col, D[lane * 16 + r]); hits++; }
      }
      printf("  half %d hits %d\n", lb_half, hits);
    }
    la = 37; ba = 3;
    for (int lb_half = 0; lb_half < 2; ++lb_half)
      for (int bb = 0; bb < 32; ++bb) {
        std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0); A[la * 32 + ba] = 0x38;
        int lb = 9 + 32 * lb_half; B[lb * 32 + bb] = 0x40;
        run("");
        for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) if (D[lane * 16 + r] != 0) { int row, col; outrc(lane, r, row, col); printf("  A(l%d,b%d) x B(l%d,b%d) -> D[%d][%d] = %g\n", la, ba, lb, bb, row, col, D[lane * 16 + r]); }
      }
  }
  // Test 3: scales.  A all ones, B all ones -> D = 64 with scale 1.  Set sa of ONE lane to 128 (x2), see which outputs change.
  for (int t = 0; t < 4; ++t) {
    std::fill(A.begin(), A.end(), 0x38); std::fill(B.begin(), B.end(), 0x38); std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
    int ls = (t & 1) ? 40 : 8;
    if (t & 2) sb[ls] = 128; else sa[ls] = 128;
    run(t & 2 ? "T3: scale_b of one lane = 2" : "T3: scale_a of one lane = 2");
    int changed = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) if (D[lane * 16 + r] != 64.f) { int row, col; outrc(lane, r, row, col); if (changed < 6) printf("  lane_s %d: D[%d][%d] = %g\n", ls, row, col, D[lane * 16 + r]); changed++; }
    printf("  changed %d outputs\n", changed);
  }
  // Test 4: scale byte selection: put different bytes in the scale word
  {
    std::fill(A.begin(), A.end(), 0x38); std::fill(B.begin(), B.end(), 0x38); std::fill(sb.begin(), sb.end(), 127);
    for (auto& v : sa) v = 127 | (129 << 8) | (131 << 16) | (133 << 24);
    run("T4: scale_a word = bytes {127,129,131,133}, opsel 0");
    printf("  D[0][0] = %g (64 => byte0)\n", D[0]);
  }
  return 0;
}
