// Micro-benchmark: LDS read-modify-write throughput on gfx950 (float atomic vs int atomic vs plain RMW), used to
// size the LDS-privatised scatter of the deformable-conv grad_input kernel.  Indices are generated in registers (no
// global loads in the loop); 2 workgroups per CU.   Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NREG = 8192;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float reg[NREG];
  for (int i = threadIdx.x; i < NREG; i += 256) reg[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4, wave = threadIdx.x >> 6;
  float v = 1.0f + lane;
  unsigned h = 1234567u + wave * 977u + lg * 131u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      const int e = (h >> 10) % 440;                       // voxel slot, one per 16-lane group
      if (MODE == 0) atomicAdd(&reg[e * 17 + l15], v);        // float atomic, 4 voxels x 16 channels (stride 17)
      if (MODE == 1) atomicAdd((int*)&reg[e * 17 + l15], 3);  // int atomic, same addresses
      if (MODE == 2) reg[e * 17 + l15] += v;                  // plain RMW, same addresses (racy, timing only)
      if (MODE == 3) atomicAdd(&reg[(e & 127) * 64 + lane], v);   // float atomic, 64 consecutive floats
      if (MODE == 4) atomicAdd(&reg[(lane * 33 + u * 7) & (NREG - 1)], v);   // float atomic, conflict-free fixed pattern
      if (MODE == 5) atomicAdd((double*)&reg[2 * (e * 9 + (l15 & 7))], (double)v);               // f64 atomic, 8 ch (stride 9 doubles)
      if (MODE == 6) atomicAdd((unsigned long long*)&reg[2 * (e * 9 + (l15 & 7))], 3ull);        // u64 atomic
      if (MODE == 7) atomicAdd((unsigned long long*)&reg[2 * ((e & 255) * 16 + l15)], 3ull);     // u64 atomic, 16 ch (stride 16)
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < NREG; i += 256) s += reg[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* dout) {
  const int iters = 4000, blocks = 512;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(dout, 10);
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(dout, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double lane_ops = (double)blocks * 256 * iters * 8;
  printf("%-44s %8.3f ms   %8.1f G lane-ops/s   %.2f lanes/clk/CU @2.4GHz\n", name, ms, lane_ops / ms / 1e6, lane_ops / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
  float* dout;
  hipMalloc(&dout, 512 * 256 * 4);
  run<0>("ds_add_f32  4 voxels x 16 ch (stride 17)", dout);
  run<1>("ds_add_u32  4 voxels x 16 ch (stride 17)", dout);
  run<2>("plain RMW   4 voxels x 16 ch (stride 17)", dout);
  run<3>("ds_add_f32  64 consecutive floats", dout);
  run<4>("ds_add_f32  conflict-free pattern", dout);
  run<5>("ds_add_f64  4 voxels x 8 ch (stride 9)", dout);
  run<6>("ds_add_u64  4 voxels x 8 ch (stride 9)", dout);
  run<7>("ds_add_u64  4 voxels x 16 ch (stride 16)", dout);
  return 0;
}
