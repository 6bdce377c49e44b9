// Probe (diagnostics): operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 A and B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  memcpy(&a, A + l * 32, 32);
  memcpy(&b, B + l * 32, 32);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int k = 0; k < 4; ++k) D[l * 4 + k] = c[k];
}

int main() {
  unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 1024);
  std::vector<unsigned char> A(2048), B(2048);
  std::vector<int> sa(64, 127), sb(64, 127);
  std::vector<float> D(256);
  auto run = [&]() {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  };
  // K pairing: A one-hot (la, ba) against B one-hot (lb, bb) with lb & 15 == 3
  const int las[] = {5, 21, 37, 53}, bas[] = {0, 15, 16, 31};
  for (int la : las) for (int ba : bas) {
    printf("A(l=%2d,b=%2d) row %2d pairs with:", la, ba, la & 15);
    for (int kg = 0; kg < 4; ++kg) for (int bb = 0; bb < 32; ++bb) {
      std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
      A[la * 32 + ba] = 0x38; const int lb = 3 + 16 * kg; B[lb * 32 + bb] = 0x40;
      run();
      for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l * 4 + r] != 0) printf("  B(l=%2d,b=%2d) -> D(l=%2d,r=%d)=%g", lb, bb, l, r, D[l * 4 + r]);
    }
    printf("\n");
  }
  // scales: all ones -> 128; one lane's scale_a (or scale_b) doubled
  for (int t = 0; t < 8; ++t) {
    std::fill(A.begin(), A.end(), 0x38); std::fill(B.begin(), B.end(), 0x38); std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
    const int ls = 5 + 16 * (t & 3);
    if (t & 4) sb[ls] = 128; else sa[ls] = 128;
    run();
    printf("%s of lane %2d doubled:", (t & 4) ? "scale_b" : "scale_a", ls);
    int shown = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l * 4 + r] != 128.f && shown++ < 6) printf("  D(l=%2d,r=%d)=%g", l, r, D[l * 4 + r]);
    int n = 0; for (float v : D) n += v != 128.f;
    printf("  [%d changed]\n", n);
  }
  // which K block does the scale of lane group s act on?  zero A's block j (lane (5, kg=j)) and double scale_a of lane (5, kg=s):
  // row 5 = 96 if s == block j's scale source (the doubled block is the zeroed one), 128 otherwise (96 + 32).
  printf("scale source: rows = zeroed block j, cols = doubled scale lane group s -> value of D row 5\n");
  for (int side = 0; side < 2; ++side) {
    printf(side ? " scale_b / B blocks:\n" : " scale_a / A blocks:\n");
    for (int j = 0; j < 4; ++j) {
      printf("  j=%d:", j);
      for (int sg = 0; sg < 4; ++sg) {
        std::fill(A.begin(), A.end(), 0x38); std::fill(B.begin(), B.end(), 0x38); std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
        if (side == 0) { for (int b = 0; b < 32; ++b) A[(5 + 16 * j) * 32 + b] = 0; sa[5 + 16 * sg] = 128; }
        else           { for (int b = 0; b < 32; ++b) B[(5 + 16 * j) * 32 + b] = 0; sb[5 + 16 * sg] = 128; }
        run();
        // row 5 (side 0): lanes 16..31 reg 1 ; col 5 (side 1): lane 5 any reg
        printf(" %g", side == 0 ? D[16 * 4 + 1] : D[5 * 4 + 0]);
      }
      printf("\n");
    }
  }
  printf("block (group pair P, byte half h) zeroed entirely; cols = doubled scale lane group s: 96 = that scale owns the block\n");
  for (int side = 0; side < 2; ++side)
    for (int P = 0; P < 2; ++P) for (int h = 0; h < 2; ++h) {
      printf("  %s P=%d h=%d:", side ? "B" : "A", P, h);
      for (int sg = 0; sg < 4; ++sg) {
        std::fill(A.begin(), A.end(), 0x38); std::fill(B.begin(), B.end(), 0x38); std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
        std::vector<unsigned char>& M = side ? B : A;
        for (int g = 2 * P; g < 2 * P + 2; ++g) for (int b = 16 * h; b < 16 * h + 16; ++b) M[(5 + 16 * g) * 32 + b] = 0;
        (side ? sb : sa)[5 + 16 * sg] = 128;
        run();
        printf(" %g", side == 0 ? D[16 * 4 + 1] : D[5 * 4 + 0]);
      }
      printf("\n");
    }
  return 0;
}
