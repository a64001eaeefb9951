// Peak rate of ds_add_u64 / ds_add_u32 with NO address arithmetic in the loop (8 precomputed addresses per lane), at 1, 2 and 4
// waves per SIMD -- tools/lds_atomic_bench.hip spends ~10 vector instructions per atomic and a single wave issues one vector
// instruction per ~8 clocks (tools/mfma_valu_probe.hip), so that bench can be issue-bound rather than LDS-bound.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  extern __shared__ __align__(16) unsigned long long reg[];   // 100 KB: one workgroup per CU
  for (int i = threadIdx.x; i < 12800; i += blockDim.x) reg[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned h = 1234567u + wave * 977u + (lane >> 3) * 131u;
  unsigned long long* a[8];
  for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u;
    const int cell = (h >> 10) % 1600;                 // one cell of 8 u64 per 8-lane group: the scatter's pattern
    a[u] = reg + cell * 8 + (lane & 7);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) atomicAdd(a[u], 3ull);
      if (MODE == 1) atomicAdd((unsigned*)a[u], 3u);
      if (MODE == 2) atomicAdd((unsigned*)a[u], 3u), atomicAdd((unsigned*)a[u] + 1, 5u);
    }
  }
  __syncthreads();
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < 12800; i += blockDim.x) s += reg[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
template <int MODE>
void run(const char* name, int waves, float* dout) {
  const int iters = 2000, blocks = 256;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), 102400, 0, dout, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), 102400, 0, dout, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double lane_ops = (double)blocks * 64 * waves * iters * 8 * (MODE == 2 ? 2 : 1);
  printf("%-28s %2d waves/CU %8.3f ms  %.2f lanes/clk/CU @2.4GHz\n", name, waves, ms, lane_ops / 256 / (ms * 1e-3 * 2.4e9));
}
int main() {
  float* dout; (void)hipMalloc(&dout, 256 * 1024 * 4);
  for (int w : {4, 8, 16}) { run<0>("ds_add_u64", w, dout); run<1>("ds_add_u32", w, dout); run<2>("2 x ds_add_u32", w, dout); }
  return 0;
}
