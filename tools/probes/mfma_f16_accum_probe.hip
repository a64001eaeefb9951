// How exactly does v_mfma_f32_32x32x16_f16 add a 16-term dot product to an accumulator that is much larger than the products?
// D = A B + C with random A (32 x 16), B (16 x 32), C (32 x 32) at chosen magnitudes; error against the exactly rounded result in ulps of D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
  const int l = threadIdx.x, l31 = l & 31, hh = l >> 5;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[l31 * 16 + 8 * hh + i]; b[i] = B[(8 * hh + i) * 32 + l31]; }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = C[((j & 3) + 8 * (j >> 2) + 4 * hh) * 32 + l31];
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int j = 0; j < 16; ++j) D[((j & 3) + 8 * (j >> 2) + 4 * hh) * 32 + l31] = acc[j];
}
static double rnd() { return (double)rand() / RAND_MAX * 2 - 1; }
int main() {
  _Float16 hA[512], hB[512]; float hC[1024], hD[1024];
  _Float16 *dA, *dB; float *dC, *dD;
  (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
  const double cases[][3] = {{1, 1, 0}, {1, 1, 16}, {0.25, 4096, 524288}, {ldexp(1, -13), 4096, 524288}, {0.25, 1, 524288}, {ldexp(1, -21), 4096, 0.09}, {ldexp(1.0, -21), 1, 0.09}, {1, 1, 1e6}, {0, 4096, 0.09}, {0, 30000, 1e-5}, {ldexp(1.0, -21), 4096, 100.0}, {ldexp(1.0, -16), 16384, 0.09}, {ldexp(1.0, -13), 16384, 0.09}, {ldexp(1.0, -21), 4096, 8.0}, {ldexp(1.0, -21), 4096, 1.0}};
  for (auto& cs : cases) {
    double worst = 0, sum = 0; int n = 0; double bias = 0;
    for (int trial = 0; trial < 50; ++trial) {
      for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)(cs[0] * rnd()); hB[i] = (_Float16)(cs[1] * rnd()); }
      for (int i = 0; i < 1024; ++i) hC[i] = (float)(cs[2] * rnd());
      (void)hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
      for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c) {
          double ex = hC[r * 32 + c];
          for (int kk = 0; kk < 16; ++kk) ex += (double)hA[r * 16 + kk] * (double)hB[kk * 32 + c];
          const double ulp = ldexp(1.0, ilogb((double)hD[r * 32 + c] != 0 ? fabs(ex) : 1e-30) - 23);
          const double err = ((double)hD[r * 32 + c] - ex) / ulp;
          worst = fmax(worst, fabs(err)); sum += fabs(err); bias += err * (ex > 0 ? 1 : -1); ++n;
        }
    }
    printf("|a| %.3g |b| %.3g |c| %.3g: worst %.2f ulp, mean |err| %.3f ulp, mean signed (toward larger magnitude +) %.3f ulp\n", cs[0], cs[1], cs[2], worst, sum / n, bias / n);
  }
  return 0;
}
