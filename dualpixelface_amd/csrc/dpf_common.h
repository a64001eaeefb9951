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

// Deterministic mode (dpf_set_deterministic(1) / DPF_DETERMINISTIC=1, layout.hip): every reduction that merges partial results with float
// atomics switches to a form whose result does not depend on the order in which workgroups retire -- one committing workgroup per output
// address, phased launches of overlapping tiles, or integer accumulation (below).  Slower; bitwise reproducible run to run.
int dpf_deterministic();

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

// Order-independent accumulation of floats: a value is added as TWO 64-bit integers (units 2^-24 and 2^-56) with integer atomics.  The
// conversion is a pure function of the value (exact for |v| >= 2^-33, rounded to 2^-56 below that; |sum| < 2^38), integer adds commute, so
// the pair -- and dpf_det_value() of it -- is the same whatever order the contributions arrive in.  acc2: two zero-initialised long longs.
__device__ __forceinline__ void dpf_det_add(long long* acc2, float v) {
  const float s = v * 16777216.f;                       // exact (power of two)
  const float t = truncf(s);
  const float r = s - t;                                // exact; 0 once |s| >= 2^23
  const long long qh = (long long)t;
  const long long ql = (long long)rintf(r * 4294967296.f);
  if (qh != 0) atomicAdd(reinterpret_cast<unsigned long long*>(acc2), (unsigned long long)qh);
  if (ql != 0) atomicAdd(reinterpret_cast<unsigned long long*>(acc2) + 1, (unsigned long long)ql);
}
__device__ __forceinline__ float dpf_det_value(const long long* acc2) {
  return (float)((double)acc2[0] * 5.9604644775390625e-08 + (double)acc2[1] * 1.3877787807814457e-17);   // 2^-24, 2^-56
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
