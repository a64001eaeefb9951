// Is a product with a SUBNORMAL f16 input exact in v_mfma_f32_32x32x16_f16?  D = a b (one non-zero k), a = m 2^-24, b normal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A, const _Float16* B, float* D, int swap) {
  const int l = threadIdx.x, l31 = l & 31, hh = l >> 5;
  f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hh == 0) { a[0] = A[l31]; b[0] = B[l31]; }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  acc = swap ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int j = 0; j < 16; ++j) D[((j & 3) + 8 * (j >> 2) + 4 * hh) * 32 + l31] = acc[j];
}
int main() {
  _Float16 hA[32], hB[32]; float hD[1024];
  _Float16 *dA, *dB; float* dD;
  (void)hipMalloc(&dA, 64); (void)hipMalloc(&dB, 64); (void)hipMalloc(&dD, 4096);
  for (int swap = 0; swap < 2; ++swap)
    for (int be = -12; be <= 14; be += 2) {
      int bad = 0, n = 0; double worst = 0;
      for (int trial = 0; trial < 20; ++trial) {
        for (int i = 0; i < 32; ++i) { hA[i] = (_Float16)ldexp((double)(1 + rand() % 1023), -24); hB[i] = (_Float16)ldexp(1.0 + (rand() % 1024) / 1024.0, be); }
        (void)hipMemcpy(dA, hA, 64, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, swap);
        (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        for (int r = 0; r < 32; ++r)
          for (int c = 0; c < 32; ++c) {
            const double ex = swap ? (double)hB[r] * (double)hA[c] : (double)hA[r] * (double)hB[c];
            const double rel = fabs((double)hD[r * 32 + c] - ex) / ex;
            if (rel != 0) ++bad;
            worst = fmax(worst, rel); ++n;
          }
      }
      printf("subnormal operand %s, other operand 2^%d: %d of %d products inexact, worst relative error 2^%.1f\n", swap ? "B" : "A", be, bad, n, worst > 0 ? log2(worst) : -99.);
    }
  return 0;
}
