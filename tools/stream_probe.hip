// What does a BatchNorm-shaped streaming kernel reach on MI355X?  y = a * x + b per row (1 read + 1 write), dx-like (2 reads + 1 write) and a
// two-stream reduction (2 reads), over [128 rows][6 291 456] fp32 (the hourglass tensor), for several grid shapes.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe && ./tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr long long S = 8LL * 256 * 384;   // per row: [4 x 32 rows][8 x 256 x 384] = the hourglass tensor (403 MB)
constexpr int ROWS = 128;

template <int MODE, int UNR, bool NT>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x, const float* __restrict__ d, float* __restrict__ y, float* __restrict__ sums, int chunk) {
  const int row = blockIdx.y;
  const long long base = (long long)row * S;
  const float a = 1.0001f + row, b = 0.5f;
  float acc = 0.f;
  for (long long s0 = (long long)blockIdx.x * chunk; s0 < S; s0 += (long long)gridDim.x * chunk) {
    const long long s1 = s0 + chunk < S ? s0 + chunk : S;
#pragma unroll UNR
    for (long long s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
      v4f xv, dv = {0, 0, 0, 0};
      if (NT) { xv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + base + s)); if (MODE) dv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(d + base + s)); }
      else { xv = *reinterpret_cast<const v4f*>(x + base + s); if (MODE) dv = *reinterpret_cast<const v4f*>(d + base + s); }
      if (MODE == 2) { acc += xv.x * dv.x + xv.y * dv.y + xv.z * dv.z + xv.w * dv.w; continue; }
      v4f o = {fmaf(xv.x, a, b) + dv.x, fmaf(xv.y, a, b) + dv.y, fmaf(xv.z, a, b) + dv.z, fmaf(xv.w, a, b) + dv.w};
      if (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(y + base + s)); else *reinterpret_cast<v4f*>(y + base + s) = o;
    }
  }
  if (MODE == 2) {
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[row], acc);
  }
}

template <int MODE, int UNR, bool NT>
void run(const char* what, int gx, int chunk, const float* x, const float* d, float* y, float* sums) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rows<MODE, UNR, NT>), dim3(gx, ROWS), dim3(256), 0, 0, x, d, y, sums, chunk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double bytes = (double)ROWS * S * 4 * (MODE == 0 ? 2 : (MODE == 1 ? 3 : 2));
  printf("%-44s grid.x %5d chunk %6d unroll %d %s  %.3f ms  %.2f TB/s\n", what, gx, chunk, UNR, NT ? "nt" : "  ", best, bytes / best / 1e9);
}

int main() {
  float *x, *d, *y, *sums;
  const size_t n = (size_t)ROWS * S;
  hipMalloc(&x, n * 4); hipMalloc(&d, n * 4); hipMalloc(&y, n * 4); hipMalloc(&sums, ROWS * 4);
  hipMemset(x, 0, n * 4); hipMemset(d, 0, n * 4); hipMemset(sums, 0, ROWS * 4);
  const int full = (int)((S + 4095) / 4096);
  run<0, 1, false>("1R1W one-shot blocks (current shape)", full, 4096, x, d, y, sums);
  run<0, 4, false>("1R1W one-shot blocks, unroll 4", full, 4096, x, d, y, sums);
  run<0, 4, false>("1R1W 16 blocks per row, chunk 4096 strided", 16, 4096, x, d, y, sums);
  run<0, 4, false>("1R1W 32 blocks per row, chunk 4096 strided", 32, 4096, x, d, y, sums);
  run<0, 4, false>("1R1W 64 blocks per row, chunk 4096 strided", 64, 4096, x, d, y, sums);
  run<0, 4, true>("1R1W 32 blocks per row, nontemporal", 32, 4096, x, d, y, sums);
  run<0, 4, true>("1R1W one-shot blocks, nontemporal", full, 4096, x, d, y, sums);
  run<1, 1, false>("2R1W one-shot blocks (current shape)", full, 4096, x, d, y, sums);
  run<1, 4, false>("2R1W 32 blocks per row", 32, 4096, x, d, y, sums);
  run<1, 4, true>("2R1W 32 blocks per row, nontemporal", 32, 4096, x, d, y, sums);
  run<1, 4, true>("2R1W one-shot, nontemporal", full, 4096, x, d, y, sums);
  run<2, 2, false>("2R reduce, 96 blocks per row of 65536 (current)", 96, 65536, x, d, y, sums);
  run<2, 4, false>("2R reduce, 32 blocks per row, chunk 4096 strided", 32, 4096, x, d, y, sums);
  run<2, 4, true>("2R reduce, 32 blocks per row, nontemporal", 32, 4096, x, d, y, sums);
  run<2, 4, true>("2R reduce, 96 x 65536, nontemporal", 96, 65536, x, d, y, sums);
  return 0;
}
