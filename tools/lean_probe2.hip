// Within ONE wave (one wave per SIMD): how much vector-ALU / LDS work fits in the shadow of an fp32 MFMA?  Loop body = 1 MFMA + NV v_pk_fma_f32
// or + NL ds_read_b128, program order interleaved.  Prints clocks per loop body.
//   hipcc --offload-arch=gfx950 -O3 tools/lean_probe2.hip -o tools/lean_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND /*0: 32x32x2, 1: 16x16x4*/, int NV, int NL>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
  extern __shared__ __align__(16) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0};
  f32x4 b0 = {0}, b1 = {0};
  f32x2 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)i, 1.f};
  f32x2 w = {1.0001f, 0.9999f}, v = {0.5f, 0.25f};
  float fa = 1.0001f * lane, fb = 0.5f;
  const unsigned base = (unsigned)(lane * 16);
  f32x4 r[8];
  for (int i = 0; i < 8; ++i) r[i] = f32x4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (KIND == 0) {
        if (m & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(fa), "v"(fb));
      } else if (KIND == 4) {
        if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(r[6]), "v"(r[7]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(r[6]), "v"(r[7]));
      } else if (KIND == 3) {
        if (m & 1) asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(w), "v"(v));
        else asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(w), "v"(v));
      } else if (KIND == 1) {
        if (m & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b0) : "v"(fa), "v"(fb));
      }
#pragma unroll
      for (int j = 0; j < (NV >= 100 ? 0 : NV); ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j & 7]) : "v"(w), "v"(v));
#pragma unroll
      for (int j = 0; j < (NV >= 100 ? NV - 100 : 0); ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j & 7].x) : "v"(fa), "v"(fb));
#pragma unroll
      for (int j = 0; j < NL; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j & 7]) : "v"(base), "n"((j & 7) * 1024));
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_nop 15\n s_nop 15");
  const long long t1 = __builtin_readcyclecounter();
  float s = a0[0] + a1[1] + b0[0] + b1[1];
  for (int i = 0; i < 8; ++i) s += acc[i].x + r[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = t1 - t0;
}

template <int KIND, int NV, int NL>
void run(float* out, long long* clk) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<KIND, NV, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<KIND, NV, NL>), dim3(256), dim3(256), 100 * 1024, 0, out, clk, iters); hipDeviceSynchronize(); }
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-13s + %3d v_pk_fma (>=100: v_fma_f32 count + 100) + %2d ds_read_b128 per MFMA: %7.1f clk per MFMA\n", KIND == 0 ? "32x32x2" : (KIND == 1 ? "16x16x4" : (KIND == 3 ? "32x32x8 bf16" : (KIND == 4 ? "32x32x16 bf16" : "no MFMA"))), NV, NL, (double)c / iters / 4);
}

int main() {
  float* out; long long* clk; hipMalloc(&out, 4 * 256 * 256); hipMalloc(&clk, 8);
  run<0, 0, 0>(out, clk); run<0, 4, 0>(out, clk); run<0, 8, 0>(out, clk); run<0, 12, 0>(out, clk); run<0, 16, 0>(out, clk);
  run<0, 0, 2>(out, clk); run<0, 0, 4>(out, clk); run<0, 0, 8>(out, clk); run<0, 8, 4>(out, clk);
  run<2, 8, 0>(out, clk); run<2, 16, 0>(out, clk); run<2, 0, 4>(out, clk); run<2, 0, 8>(out, clk); run<2, 8, 4>(out, clk);
  run<4, 0, 0>(out, clk); run<4, 102, 0>(out, clk); run<4, 104, 0>(out, clk); run<4, 108, 0>(out, clk); run<4, 2, 0>(out, clk); run<4, 4, 0>(out, clk); run<4, 104, 2>(out, clk);
  run<0, 104, 0>(out, clk); run<0, 108, 0>(out, clk); run<3, 104, 0>(out, clk);
  run<3, 0, 0>(out, clk); run<3, 2, 0>(out, clk); run<3, 4, 0>(out, clk); run<3, 8, 0>(out, clk); run<3, 0, 2>(out, clk); run<3, 2, 1>(out, clk); run<3, 3, 1>(out, clk);
  run<1, 0, 0>(out, clk); run<1, 4, 0>(out, clk); run<1, 8, 0>(out, clk); run<1, 0, 2>(out, clk); run<1, 0, 4>(out, clk); run<1, 4, 2>(out, clk);
  return 0;
}
