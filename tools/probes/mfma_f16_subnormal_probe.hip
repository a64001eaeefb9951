// How v_mfma_f32_32x32x16_f16 treats subnormal f16 inputs and small products next to a large accumulator (gfx950).
// hipcc --offload-arch=gfx950 -O2 -o mfma_f16_subnormal_probe mfma_f16_subnormal_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// every lane: A row = a (all 8 k-values of both halves), B col = b, C = c  ->  D = c + 16 a b
__global__ void probe(const float* a, const float* b, const float* c, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    f16x8 av, bv;
    for (int k = 0; k < 8; ++k) { av[k] = (_Float16)a[i]; bv[k] = (_Float16)b[i]; }
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = c[i];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[i] = acc[0];
  }
}
// one product only (k = 0 of lane half 0), the rest zero
__global__ void probe1(const float* a, const float* b, const float* c, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    f16x8 av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((threadIdx.x >> 5) == 0) { av[0] = (_Float16)a[i]; bv[0] = (_Float16)b[i]; }
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = c[i];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[i] = acc[0];
  }
}
int main() {
  const int n = 20;
  //                 subnormal a        subnormal b     both normal        small product + big C ...
  float ha[n] = {ldexpf(1, -20), 1024.f,          3.f,      ldexpf(1, -13), ldexpf(1, -13), ldexpf(1, -13), ldexpf(1,-13), ldexpf(3,-16), 1.f, 1.f, 1.f, ldexpf(1,-24), ldexpf(3,-24), ldexpf(3,-24), ldexpf(3,-24), ldexpf(5,-24), 0.01f, 0.01f, ldexpf(3,-24), ldexpf(3,-14)};
  float hb[n] = {1024.f,         ldexpf(1, -20),  5.f,      4096.f,         4096.f,         4096.f,         4096.f,        4096.f,        ldexpf(1,-10), ldexpf(1,-10), ldexpf(1,-10), 16384.f, 0.01f, 0.01f, 1.5f, 0.3f, ldexpf(3,-24), ldexpf(3,-20), ldexpf(3,-24), ldexpf(3,-14)};
  float hc[n] = {0.f,            0.f,             0.f,      0.f,            1024.f,         ldexpf(1, 19),  ldexpf(1, 22), ldexpf(1,19),  ldexpf(1,12), ldexpf(1,13), ldexpf(1,14), 0.f, 0.f, ldexpf(1,-6), 0.f, ldexpf(1,-9), 0.f, ldexpf(1,-9), 0.f, 0.f};
  float *da, *db, *dc, *dout, ho[n], ho1[n];
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, hc, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc, dout, n);
  hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(probe1, dim3(1), dim3(64), 0, 0, da, db, dc, dout, n);
  hipMemcpy(ho1, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) {
    const double ex16 = (double)hc[i] + 16.0 * (double)ha[i] * hb[i], ex1 = (double)hc[i] + (double)ha[i] * hb[i];
    printf("a %.6g b %.6g c %.6g | 16 products: got %.10g exact %.10g (fp32-rounded %.10g) | 1 product: got %.10g exact %.10g (fp32-rounded %.10g)\n", ha[i], hb[i], hc[i], ho[i], ex16,
           (double)(float)ex16, ho1[i], ex1, (double)(float)ex1);
  }
  return 0;
}
