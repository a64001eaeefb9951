// Sustained shader clock under load: s_memtime ticks vs wall time for (a) a dependent VALU chain, (b) back-to-back fp32 MFMA on all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void valu_k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) a = fmaf(a, b, 1e-7f);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void mfma_k(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 0.5f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 4 * 256 * 4096); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      int iters = mode == 0 ? 20000 : 40000;
      int blocks = 256 * (mode == 0 ? 8 : 8);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(valu_k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(mfma_k, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      if (mode == 0) printf("valu: %.2f ms, block0 ticks %lld, ops/wave %lld\n", ms, c, (long long)iters * 64);
      else {
        double flops = (double)blocks * 4 * iters * 16 * 4096.0;
        printf("mfma: %.2f ms, block0 ticks %lld, %.1f TFLOP/s, cycles per mfma per wave (if 2 blocks/CU share a SIMD) %.1f\n", ms, c, flops / ms * 1e-9, (double)c / (iters * 16.0));
      }
    }
  }
  return 0;
}
