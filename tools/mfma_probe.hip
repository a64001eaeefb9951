// What fp32 MFMA rate survives realistic operand feeding?  Variants of a 32x32x2 f32 loop at 4 waves/SIMD (4 blocks x 4 waves / CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ __launch_bounds__(256, 4) void k(const float* __restrict__ w, float* out, int iters) {
  __shared__ float lds[7424];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  for (int i = tid; i < 7424; i += 256) lds[i] = (float)((i * 2654435761u) >> 8) * 1e-9f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = w[tid], b[NACC];
  for (int t = 0; t < NACC; ++t) b[t] = lds[wave * 34 * NACC + t * 34 + l31 + hh * 1836];
  for (int it = 0; it < iters; ++it) {
    const int off = (it * 37) & 1023;
    float bn[NACC], an = a;
    if (MODE >= 1) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) bn[t] = lds[wave * 34 * NACC + t * 34 + l31 + hh * 1836 + off];
    }
    if (MODE >= 2) an = w[((it * 64) & 16383) + lane];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
    if (MODE >= 1) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) b[t] = bn[t];
    }
    a = an;
    if (MODE >= 3 && (it & 31) == 31) __syncthreads();
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE, int NACC>
void run(const char* name, const float* w, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 1024 * 2;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(blocks), dim3(256), 0, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-44s %.2f ms  %.1f TFLOP/s\n", name, ms, (double)blocks * 4 * iters * NACC * 4096.0 / ms * 1e-9);
  }
}
int main() {
  float *w, *out; hipMalloc(&w, 4 * 65536); hipMalloc(&out, 4 * 256 * 4096);
  float* h = (float*)malloc(4 * 65536); for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(w, h, 4 * 65536, hipMemcpyHostToDevice);
  run<0, 4>("regs only, 4 acc", w, out);
  run<1, 4>("B from LDS (1 ahead), 4 acc", w, out);
  run<2, 4>("B from LDS + A from L2, 4 acc", w, out);
  run<3, 4>("... + barrier every 32 iters", w, out);
  run<0, 8>("regs only, 8 acc", w, out);
  run<1, 8>("B from LDS (1 ahead), 8 acc", w, out);
  run<2, 8>("B from LDS + A from L2, 8 acc", w, out);
  run<1, 2>("B from LDS (1 ahead), 2 acc", w, out);
  return 0;
}
