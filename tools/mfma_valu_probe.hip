// Does vector-ALU work of the same SIMD overlap a v_mfma_f32_16x16x4_f32 stream?  One loop body = NM MFMAs (two independent
// accumulator chains) + NV independent v_fma_f32, at 1 or 2 waves per SIMD.  Prints core clocks per loop body (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, bool INTER>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters, int mfma_waves_mask) {
  extern __shared__ float pad[];
  const int wave = threadIdx.x >> 6;
  const bool do_m = (mfma_waves_mask >> (wave >> 2)) & 1;      // wave / 4 = which of the SIMD's resident waves this is
  const bool do_v = (mfma_waves_mask >> (2 + (wave >> 2))) & 1;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x[8], fa = 1.0001f * (threadIdx.x + 1), fb = 0.5f;
  for (int i = 0; i < 8; ++i) x[i] = i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (INTER) {      // every wave: NM x (1 MFMA + NV / NM VALU)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (m & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(fa), "v"(fb));
#pragma unroll
        for (int v = 0; v < NV / (NM > 0 ? NM : 1); ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[(m * (NV / (NM > 0 ? NM : 1)) + v) & 7]) : "v"(fa), "v"(fb));
      }
    } else {
      if (do_m)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          if (m & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(fa), "v"(fb));
          else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(fa), "v"(fb));
        }
      if (do_v)
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v & 7]) : "v"(fa), "v"(fb));
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  const long long t1 = __builtin_readcyclecounter();
  float s = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NM, int NV, bool INTER>
void run(const char* what, int waves, int mask, float* out, long long* clk) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<NM, NV, INTER>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, INTER>), dim3(256), dim3(64 * waves), 100 * 1024, 0, out, clk, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-64s NM %2d NV %3d  %7.1f clk/body   (%.3f ms)\n", what, NM, NV, (double)c / iters, ms);
}

int main() {
  float* out; long long* clk; hipMalloc(&out, 4 * 512 * 256); hipMalloc(&clk, 8);
  // mask bits: 0 = first wave of a SIMD runs the MFMAs, 1 = second wave runs MFMAs, 2 = first wave runs VALU, 3 = second wave runs VALU
  run<8, 0, false>("1 wave/SIMD: MFMA only", 4, 0x1, out, clk);
  run<0, 48, false>("1 wave/SIMD: VALU only", 4, 0x4, out, clk);
  run<8, 48, false>("1 wave/SIMD: 8 MFMA then 48 VALU", 4, 0x5, out, clk);
  run<8, 48, true>("1 wave/SIMD: interleaved 1 MFMA : 6 VALU", 4, 0x5, out, clk);
  run<8, 24, true>("1 wave/SIMD: interleaved 1 MFMA : 3 VALU", 4, 0x5, out, clk);
  run<8, 96, true>("1 wave/SIMD: interleaved 1 MFMA : 12 VALU", 4, 0x5, out, clk);
  run<8, 48, false>("2 waves/SIMD: wave A MFMA, wave B VALU", 8, 0x1 | 0x8, out, clk);
  run<8, 48, false>("2 waves/SIMD: both MFMA then VALU", 8, 0xF, out, clk);
  run<8, 48, true>("2 waves/SIMD: both interleaved 1:6", 8, 0xF, out, clk);
  run<8, 0, false>("2 waves/SIMD: both MFMA only", 8, 0x3, out, clk);
  run<0, 48, false>("2 waves/SIMD: both VALU only", 8, 0xC, out, clk);
  return 0;
}
