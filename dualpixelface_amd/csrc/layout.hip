// Pure data-movement kernels that keep the model's tensor plumbing off eager PyTorch: channel-range copies
// (torch.cat / torch.stack / slicing in the reference: src/model/stereodpnet/modules.py:42,129, mainmodel.py:98-100),
// the [B,C,D,S] <-> [B,D,C,S] permutation in front of the shared 2-D normal convs (normal_module.py:185,187), the channel
// maximum returned as `ref_feature` (mainmodel.py:104) and the closed-form replay of the attention BatchNorm's running
// statistics.  All HBM-bound, float4 where the row length allows.
#include "dpf_common.h"
#include <cstdlib>

// ---- deterministic mode (process-wide; see dpf_common.h)
namespace { int g_deterministic = -1; }
int dpf_deterministic() {
  if (g_deterministic < 0) g_deterministic = getenv("DPF_DETERMINISTIC") ? (atoi(getenv("DPF_DETERMINISTIC")) != 0) : 0;
  return g_deterministic;
}
extern "C" int dpf_set_deterministic(int on) {
  g_deterministic = on ? 1 : 0;
  return DPF_OK;
}
extern "C" int dpf_get_deterministic(void) { return dpf_deterministic(); }

// ---- measurement aid (bench.py): the shader clock the chip holds while other streams keep it busy.  One lane records the shader cycle
// counter and the constant 100 MHz counter, naps until `spin_us` microseconds have passed, records both again: out4 = {cycles0, ticks0,
// cycles1, ticks1}; MHz = (cycles1 - cycles0) / (ticks1 - ticks0) * 100.
__global__ void dpf_clock_probe_kernel(unsigned long long* __restrict__ out4, unsigned long long spin_ticks) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  while (r1 - r0 < spin_ticks) {
    __builtin_amdgcn_s_sleep(64);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  out4[0] = c0; out4[1] = r0; out4[2] = c1; out4[3] = r1;
}
extern "C" int dpf_debug_clock_probe(unsigned long long* out4, int spin_us, void* stream) {
  dpf_clear_error();
  if (!out4 || spin_us <= 0 || spin_us > 2000000) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(dpf_clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out4, (unsigned long long)spin_us * 100ULL);
  return dpf_check_launch();
}

namespace {

// dst[n, cd0 + c, s] = src[n, cs0 + c, s]   for c < ncopy;  src has Cs channels, dst has Cd.
// grid.y = sample n (the ncopy * S elements of a sample are contiguous on both sides): no integer division per element
__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Cs, int cs0, int Cd, int cd0, int ncopy,
                                     long long S, int accumulate) {
  const long long per = (long long)ncopy * S;
  const int n = blockIdx.y;
  const float* sp = src + ((long long)n * Cs + cs0) * S;
  float* dp = dst + ((long long)n * Cd + cd0) * S;
  if ((S & 3) == 0) {
    const long long per4 = per >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += (long long)gridDim.x * blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(sp)[i];
      float4* d = reinterpret_cast<float4*>(dp) + i;
      if (accumulate) {
        float4 o = *d;
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *d = o;
      } else {
        *d = v;
      }
    }
  } else {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long long)gridDim.x * blockDim.x) {
      const float v = sp[e];
      dp[e] = accumulate ? dp[e] + v : v;
    }
  }
}

// dst[b, d, c, s] = src[b, c, d, s]   (A = C, Bd = D)  -- the same kernel inverts itself with A and Bd swapped.
// grid.y = (b, d, c) row of S contiguous elements on both sides
__global__ void swap_axes_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int A, int Bd, long long S) {
  const int row = blockIdx.y;                 // dst order [b][d][c]
  const int c = row % A;
  const int d = (row / A) % Bd;
  const long long b = row / (A * Bd);
  const float* sp = src + ((b * A + c) * Bd + d) * S;
  float* dp = dst + (long long)row * S;
  if ((S & 3) == 0) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (S >> 2); i += (long long)gridDim.x * blockDim.x)
      reinterpret_cast<float4*>(dp)[i] = reinterpret_cast<const float4*>(sp)[i];
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (long long)gridDim.x * blockDim.x) dp[i] = sp[i];
  }
}

__global__ void channel_max_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long long S) {
  const long long total = (long long)N * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / S, s = i - n * S;
    float m = x[(n * C) * S + s];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[(n * C + c) * S + s]);
    y[i] = m;
  }
}

// r <- keep^(2L) r + keep*G*a_f + G*a_b  (a_* = momentum * batch statistic of the ref / target call), G = sum_{j<L} keep^(2j)
__global__ void bn_replay_kernel(float* __restrict__ r, const float* __restrict__ a_f, const float* __restrict__ a_b, int C, float decay,
                                 float cf, float cb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) r[c] = decay * r[c] + cf * a_f[c] + cb * a_b[c];
}

}  // namespace

extern "C" {

int dpf_copy_channels(const float* src, float* dst, int N, int Cs, int cs0, int Cd, int cd0, int ncopy, long long S, int accumulate,
                      void* stream) {
  dpf_clear_error();
  if (!src || !dst || N <= 0 || ncopy <= 0 || cs0 < 0 || cd0 < 0 || cs0 + ncopy > Cs || cd0 + ncopy > Cd || S <= 0) return DPF_ERR_INVALID_ARG;
  if (N > 65535) return DPF_ERR_UNSUPPORTED;
  int gx = dpf_ew_grid((long long)ncopy * S / 4 + 1);
  if (gx > 2048 / N + 1) gx = 2048 / N + 1;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, src, dst, N, Cs, cs0, Cd, cd0, ncopy, S, accumulate);
  return dpf_check_launch();
}

// src [B, A, Bd, S] -> dst [B, Bd, A, S]
int dpf_swap_axes(const float* src, float* dst, int B, int A, int Bd, long long S, void* stream) {
  dpf_clear_error();
  if (!src || !dst || B <= 0 || A <= 0 || Bd <= 0 || S <= 0) return DPF_ERR_INVALID_ARG;
  if ((long long)B * A * Bd > 65535) return DPF_ERR_UNSUPPORTED;
  const int rows = B * A * Bd;
  int gx = dpf_ew_grid(S / 4 + 1);
  if (gx > 4096 / rows + 1) gx = 4096 / rows + 1;
  hipLaunchKernelGGL(swap_axes_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, src, dst, B, A, Bd, S);
  return dpf_check_launch();
}

// x [N, C, S] -> y [N, S]
int dpf_channel_max(const float* x, float* y, int N, int C, long long S, void* stream) {
  dpf_clear_error();
  if (!x || !y || N <= 0 || C <= 0 || S <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(channel_max_kernel, dim3(dpf_ew_grid((long long)N * S)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, S);
  return dpf_check_launch();
}

int dpf_bn_replay(float* running, const float* a_f, const float* a_b, int C, float decay, float cf, float cb, void* stream) {
  dpf_clear_error();
  if (!running || !a_f || !a_b || C <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_replay_kernel, dim3(dpf_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream, running, a_f, a_b, C, decay, cf, cb);
  return dpf_check_launch();
}

}  // extern "C"
