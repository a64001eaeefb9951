// Micro-benchmark: LDS read-modify-write throughput on gfx950 (float atomic vs int atomic vs plain RMW), used to
// size the LDS-privatised scatter of the deformable-conv grad_input kernel.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, float* out, int iters, int nidx) {
  __shared__ float reg[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) reg[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
    const int e = idx[(it * 16 + wave * 4 + lg) % nidx];   // voxel slot, 4 different per wave instruction
    const int a = e * 17 + l15;
    if (MODE == 0) atomicAdd(&reg[a], v);
    if (MODE == 1) atomicAdd((int*)&reg[a], (int)v);
    if (MODE == 2) reg[a] += v;
    if (MODE == 3) atomicAdd(&reg[(e & 255) * 64 + lane], v);   // 64 consecutive floats (one voxel, 64 channels)
    if (MODE == 4) reg[(e & 255) * 64 + lane] += v;
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 16384; i += 256) s += reg[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int* didx, float* dout, int nidx) {
  const int iters = 20000, blocks = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(didx, dout, 100, nidx);
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(didx, dout, iters, nidx);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double lane_ops = (double)blocks * 256 * iters;
  printf("%-34s %8.3f ms   %7.2f G lane-RMW/s   %.2f lanes/clk/CU (2.4 GHz, 1 block/CU)\n", name, ms, lane_ops / ms / 1e6,
         lane_ops / blocks / (ms * 1e-3 * 2.4e9));
}

int main() {
  const int nidx = 4096;
  std::vector<int> h(nidx);
  unsigned s = 12345;
  for (int i = 0; i < nidx; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % 900; }
  int* didx; float* dout;
  hipMalloc(&didx, nidx * 4); hipMalloc(&dout, 256 * 256 * 4);
  hipMemcpy(didx, h.data(), nidx * 4, hipMemcpyHostToDevice);
  run<0>("ds_add_f32 4x16 stride17", didx, dout, nidx);
  run<1>("ds_add_u32 4x16 stride17", didx, dout, nidx);
  run<2>("plain RMW  4x16 stride17", didx, dout, nidx);
  run<3>("ds_add_f32 64 consecutive", didx, dout, nidx);
  run<4>("plain RMW  64 consecutive", didx, dout, nidx);
  return 0;
}
