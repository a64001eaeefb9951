// Shared device/host helpers for the dpf_* HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPF_OK 0
#define DPF_ERR_INVALID_ARG (-1)
#define DPF_ERR_LAUNCH (-2)
#define DPF_ERR_UNSUPPORTED (-3)

#define DPF_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline void dpf_clear_error() { (void)hipGetLastError(); }

static inline int dpf_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DPF_OK : DPF_ERR_LAUNCH;
}

static inline int dpf_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// grid size for a grid-stride elementwise kernel (guide: cap at ~8 blocks/CU x 256 CUs)
static inline int dpf_ew_grid(long long n, int block = 256) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float dpf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float dpf_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float dpf_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float dpf_block_sum_256(float v, float* sm4) {
  v = dpf_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[w] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}

// activation codes shared by the norm/act kernels and the host side
enum { DPF_ACT_NONE = 0, DPF_ACT_RELU = 1, DPF_ACT_PRELU = 2, DPF_ACT_LEAKY = 3, DPF_ACT_SIGMOID = 4 };

__device__ __forceinline__ float dpf_act(float z, int act, float slope) {
  switch (act) {
    case DPF_ACT_RELU: return z > 0.f ? z : 0.f;
    case DPF_ACT_PRELU:
    case DPF_ACT_LEAKY: return z > 0.f ? z : z * slope;
    case DPF_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    default: return z;
  }
}
// d act(z) / dz
__device__ __forceinline__ float dpf_dact(float z, int act, float slope) {
  switch (act) {
    case DPF_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case DPF_ACT_PRELU:
    case DPF_ACT_LEAKY: return z > 0.f ? 1.f : slope;
    case DPF_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
    default: return 1.f;
  }
}
