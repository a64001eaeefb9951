// Batch / instance normalisation with fused activation and residual adds, plus stand-alone activations
// and per-channel reductions.  HBM-bound row kernels over the [N, C, S] view of an NCHW / NCDHW tensor:
// one workgroup streams a chunk of one (n, c) row with 16-byte accesses, so no per-element index division.
//
// Replaces nn.BatchNorm2d/3d + ReLU/PReLU/LeakyReLU/Sigmoid + the residual adds of the reference
// (src/module/asm/basics.py:17-58, src/model/stereodpnet/modules.py:26-52,241-260,310-325,
// src/module/asm/asm.py:138-146 InstanceNorm3d, normal_module.py:14-19,48-51).
//
//   z = (x - mean[c]) * invstd[c] * w[c % wmod] + b[c % wmod] + res        y = act(z) + res2
#include "dpf_common.h"

namespace {

constexpr int ROW_CHUNK = 4096;   // elements of one row handled by one block

// ---------------------------------------------------------------- statistics (training)
// DET (deterministic mode, dpf_common.h): grid = (1, C) -- ONE workgroup per channel walks the rows c, c + C, ... of every sample over the
// whole S, so each output address receives exactly one (partner-less) atomic add and the result does not depend on the retirement order
// of workgroups.  Its long per-thread chains accumulate in fp64.
template <bool DET> struct RedAcc { typedef float type; };
template <> struct RedAcc<true> { typedef double type; };

__device__ __forceinline__ float red_block_sum(float v, float* sm4) { return dpf_block_sum_256(v, sm4); }
__device__ __forceinline__ float red_block_sum(double v, float* sm4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  double* smd = reinterpret_cast<double*>(sm4);        // callers declare __shared__ float sm[8] (4 doubles)
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smd[w] = v;
  __syncthreads();
  return (float)((smd[0] + smd[1]) + (smd[2] + smd[3]));
}

// sums[c] = { sum(x - K_c), sum((x - K_c)^2) } with the shift K_c = x[0, c, 0] (single pass, cancellation-safe)
template <bool DET>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ sums, int N, int C, long long S,
                                                       int chunk) {
  __shared__ float sm[8];
  typedef typename RedAcc<DET>::type acc_t;
  const int c = blockIdx.y % C;
  const float K = x[(long long)c * S];
  acc_t a = 0, b = 0;
  for (int row = blockIdx.y; row < (DET ? N * C : blockIdx.y + 1); row += C) {   // row = n*C + c
    const float* xr = x + (long long)row * S;
    const long long s0 = DET ? 0 : (long long)blockIdx.x * chunk;
    const long long s1 = DET ? S : min(S, s0 + chunk);
    if ((S & 3) == 0) {
#pragma unroll 4
      for (long long s = s0 + 4 * threadIdx.x; s < s1; s += 4 * 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + s);
        const float d0 = v.x - K, d1 = v.y - K, d2 = v.z - K, d3 = v.w - K;
        a += (d0 + d1) + (d2 + d3);
        b += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    } else {
      for (long long s = s0 + threadIdx.x; s < s1; s += 256) {
        const float d = xr[s] - K;
        a += d;
        b += d * d;
      }
    }
  }
  const float af = red_block_sum(a, sm);
  const float bf = red_block_sum(b, sm);
  if (threadIdx.x == 0) {
    atomicAdd(&sums[2 * c], af);
    atomicAdd(&sums[2 * c + 1], bf);
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ sums, int C, long long S, double count,
                                   float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double K = x[(long long)c * S];
  const double m1 = sums[2 * c] / count;
  double var = sums[2 * c + 1] / count - m1 * m1;
  if (var < 0) var = 0;
  const double mu = K + m1;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// batch statistics from the per-tile (sum, sum of squares) rows a convolution's epilogue wrote (dpf_conv_forward_stats):
// slab [parts][C][2] doubles.  Stage 1: workgroup g folds the rows g, g + G, ... -- a thread owns one channel of every (256 / C2)-th
// of those rows, so a wave reads whole contiguous rows -- and leaves one row in part2 [G][C][2].  Stage 2: one workgroup folds the G
// rows and finishes.  Every sum runs in a fixed order: bitwise reproducible.  Same outputs as bn_finalize_kernel.
constexpr int FIN_G = 64;

__global__ __launch_bounds__(256) void bn_fold_partials_kernel(const double* __restrict__ slab, int parts, int C, double* __restrict__ part2) {
  __shared__ double sm[512];
  const int C2 = 2 * C;                                   // doubles per row
  const int per = 256 / C2 > 0 ? 256 / C2 : 1;            // rows read side by side
  const int col = threadIdx.x % C2, sub = threadIdx.x / C2;
  double a = 0.0;
  if (C2 <= 256) {
    if (sub < per)
      for (int t = blockIdx.x + FIN_G * sub; t < parts; t += FIN_G * per) a += slab[(long long)t * C2 + col];
    sm[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C2) {
      double s = 0.0;
      for (int u = 0; u < per; ++u) s += sm[u * C2 + threadIdx.x];
      part2[(long long)blockIdx.x * C2 + threadIdx.x] = s;
    }
  } else {                                                // C > 128 is not produced by the conv kernel; kept for completeness
    for (int c2 = threadIdx.x; c2 < C2; c2 += 256) {
      double s = 0.0;
      for (int t = blockIdx.x; t < parts; t += FIN_G) s += slab[(long long)t * C2 + c2];
      part2[(long long)blockIdx.x * C2 + c2] = s;
    }
  }
}

// one wave per channel: lane g holds folded row g, fixed-order shuffle tree
__global__ __launch_bounds__(64) void bn_finalize_partials_kernel(const double* __restrict__ part2, int G, int C, double count, float eps,
                                                                  float momentum, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, float* __restrict__ mean,
                                                                  float* __restrict__ invstd) {
  const int c = blockIdx.x, g = threadIdx.x;
  double s1 = g < G ? part2[((long long)g * C + c) * 2] : 0.0;
  double s2 = g < G ? part2[((long long)g * C + c) * 2 + 1] : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (g != 0) return;
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// cross-rank batch statistics (SyncBatchNorm): this rank's { mean, M2 = sum (x - mean)^2 } per channel from the shifted sums
__global__ void bn_local_moments_kernel(const float* __restrict__ x, const float* __restrict__ sums, int C, long long S, double count,
                                        float* __restrict__ moments) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double K = x[(long long)c * S];
  const double m1 = sums[2 * c] / count;
  double m2 = sums[2 * c + 1] - count * m1 * m1;
  if (m2 < 0) m2 = 0;
  moments[2 * c] = (float)(K + m1);
  moments[2 * c + 1] = (float)m2;
}

// pairwise-merge (Chan et al.) of W ranks' { mean, M2 } with their element counts -> global mean / invstd, running statistics
__global__ void bn_merge_moments_kernel(const float* __restrict__ moments /*[W][2C]*/, const float* __restrict__ counts /*[W]*/, int W, int C,
                                        float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                        float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double n = 0, mu = 0;
  for (int r = 0; r < W; ++r) {
    n += (double)counts[r];
    mu += (double)counts[r] * (double)moments[((long long)r * C + c) * 2];
  }
  mu /= n;
  double m2 = 0;
  for (int r = 0; r < W; ++r) {
    const double d = (double)moments[((long long)r * C + c) * 2] - mu;
    m2 += (double)moments[((long long)r * C + c) * 2 + 1] + (double)counts[r] * d * d;
  }
  const double var = m2 / n;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unb = n > 1 ? m2 / (n - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
  }
}

// eval mode: mean/invstd straight from the running statistics
__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  invstd[c] = 1.0f / sqrtf(rv[c] + eps);
}


// The streaming kernels below are specialised at compile time on the activation and on which optional operands exist: with `act` as a
// run-time argument every element went through a switch (and every iteration through `res ? load : 0` selects), and the kernels ran at
// 3.1 - 3.7 TB/s where the same access shape without them reaches 5.8 - 6.6 (tools/stream_probe.hip).  Tensors that are read or written
// once go through nontemporal accesses (+5 - 10 % in that probe).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f ld4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); }
__device__ __forceinline__ void st4(float* p, v4f v) { __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }
template <int ACT>
__device__ __forceinline__ float act_t(float z, float slope) {
  if (ACT == DPF_ACT_RELU) return z > 0.f ? z : 0.f;
  if (ACT == DPF_ACT_PRELU || ACT == DPF_ACT_LEAKY) return z > 0.f ? z : z * slope;
  if (ACT == DPF_ACT_SIGMOID) return 1.f / (1.f + expf(-z));
  return z;
}
template <int ACT>
__device__ __forceinline__ float dact_t(float z, float slope) {
  if (ACT == DPF_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (ACT == DPF_ACT_PRELU || ACT == DPF_ACT_LEAKY) return z > 0.f ? 1.f : slope;
  if (ACT == DPF_ACT_SIGMOID) { const float sg = 1.f / (1.f + expf(-z)); return sg * (1.f - sg); }
  return 1.f;
}
// run F<ACT, FLAG>() for the run-time (act, flag)
#define DPF_ACT_DISPATCH(act, flag, CALL)                                                                    \
  switch (act) {                                                                                             \
    case DPF_ACT_RELU: if (flag) { CALL(DPF_ACT_RELU, true); } else { CALL(DPF_ACT_RELU, false); } break;    \
    case DPF_ACT_PRELU: if (flag) { CALL(DPF_ACT_PRELU, true); } else { CALL(DPF_ACT_PRELU, false); } break; \
    case DPF_ACT_LEAKY: if (flag) { CALL(DPF_ACT_LEAKY, true); } else { CALL(DPF_ACT_LEAKY, false); } break; \
    case DPF_ACT_SIGMOID: if (flag) { CALL(DPF_ACT_SIGMOID, true); } else { CALL(DPF_ACT_SIGMOID, false); } break; \
    default: if (flag) { CALL(DPF_ACT_NONE, true); } else { CALL(DPF_ACT_NONE, false); } break;              \
  }

// ---------------------------------------------------------------- apply
template <int ACT, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ w,
                                                       const float* __restrict__ bsh, int wmod, const float* __restrict__ res,
                                                       const float* __restrict__ res2, int act, const float* __restrict__ slope_p,
                                                       float slope_c, float* __restrict__ y, int C, long long S, int yC, int yc0) {
  const int row = blockIdx.y;
  const int c = row % C;
  // y may be a channel slice [yc0, yc0 + C) of a tensor with yC channels (the consumer's concatenation buffer)
  if (yC > 0) y += (((long long)(row / C) * yC + yc0 + c) - row) * S;
  float scale = 1.f, shift = 0.f;
  if (mean) {
    const float g = w ? w[c % wmod] : 1.f;
    const float be = bsh ? bsh[c % wmod] : 0.f;
    scale = invstd[c] * g;
    shift = be - mean[c] * scale;
  }
  const float slope = slope_p ? slope_p[0] : slope_c;
  const long long base = (long long)row * S;
  const long long s0 = (long long)blockIdx.x * ROW_CHUNK;
  const long long s1 = min(S, s0 + ROW_CHUNK);
  (void)act;
  if ((S & 3) == 0) {
#pragma unroll 2
    for (long long s = s0 + 4 * threadIdx.x; s < s1; s += 4 * 256) {
      const v4f v = ld4(x + base + s);
      v4f r = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
      if (RES) {                                        // (either or both residuals)
        if (res) r = ld4(res + base + s);
        if (res2) r2 = ld4(res2 + base + s);
      }
      v4f o;
      o.x = act_t<ACT>(fmaf(v.x, scale, shift) + r.x, slope) + r2.x;
      o.y = act_t<ACT>(fmaf(v.y, scale, shift) + r.y, slope) + r2.y;
      o.z = act_t<ACT>(fmaf(v.z, scale, shift) + r.z, slope) + r2.z;
      o.w = act_t<ACT>(fmaf(v.w, scale, shift) + r.w, slope) + r2.w;
      *reinterpret_cast<v4f*>(y + base + s) = o;        // (plain store: the consumer conv reads it next)
    }
  } else {
    for (long long s = s0 + threadIdx.x; s < s1; s += 256) {
      const float z = fmaf(x[base + s], scale, shift) + ((RES && res) ? res[base + s] : 0.f);
      y[base + s] = act_t<ACT>(z, slope) + ((RES && res2) ? res2[base + s] : 0.f);
    }
  }
}

// ---------------------------------------------------------------- backward
// sums[c] = { sum dz, sum dz*xhat, sum_{z<0} z*dy (PReLU slope gradient) }
template <int ACT, bool RES, bool DET>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ w, const float* __restrict__ bsh, int wmod,
                                                            const float* __restrict__ res, int act, const float* __restrict__ slope_p,
                                                            float slope_c, float* __restrict__ sums, int N, int C, long long S, int chunk,
                                                            int gC = 0, int gc0 = 0) {
  __shared__ float sm[8];
  typedef typename RedAcc<DET>::type acc_t;
  const int c = blockIdx.y % C;
  float mu = 0.f, is = 1.f, g = 1.f, be = 0.f;
  if (mean) {
    mu = mean[c];
    is = invstd[c];
    g = w ? w[c % wmod] : 1.f;
    be = bsh ? bsh[c % wmod] : 0.f;
  }
  const float slope = slope_p ? slope_p[0] : slope_c;
  acc_t a = 0, b = 0, sl = 0;
  auto one = [&](float xv, float rv, float d) {
    const float xh = (xv - mu) * is;
    const float z = fmaf(xh, g, be) + rv;
    const float dz = d * dact_t<ACT>(z, slope);
    a += dz;
    b += dz * xh;
    if (ACT == DPF_ACT_PRELU && z <= 0.f) sl += z * d;
  };
  (void)act;
  // DET: one workgroup per channel walks every sample's row over the whole S (see bn_stats_kernel)
  for (int row = blockIdx.y; row < (DET ? N * C : blockIdx.y + 1); row += C) {
    const float* dyr = dy + (gC > 0 ? (((long long)(row / C) * gC + gc0 + c) - row) * S : 0);   // dy: a channel slice of a [N, gC, S] tensor
    const long long base = (long long)row * S;
    const long long s0 = DET ? 0 : (long long)blockIdx.x * chunk;
    const long long s1 = DET ? S : min(S, s0 + chunk);
    if ((S & 3) == 0) {
#pragma unroll 4
      for (long long s = s0 + 4 * threadIdx.x; s < s1; s += 4 * 256) {
        const v4f xv = *reinterpret_cast<const v4f*>(x + base + s);      // (x and dy are read again by the apply kernel: plain loads)
        const v4f dv = *reinterpret_cast<const v4f*>(dyr + base + s);
        v4f rv = {0.f, 0.f, 0.f, 0.f};
        if (RES) rv = *reinterpret_cast<const v4f*>(res + base + s);
        one(xv.x, rv.x, dv.x); one(xv.y, rv.y, dv.y); one(xv.z, rv.z, dv.z); one(xv.w, rv.w, dv.w);
      }
    } else {
      for (long long s = s0 + threadIdx.x; s < s1; s += 256) one(x[base + s], RES ? res[base + s] : 0.f, dyr[base + s]);
    }
  }
  const float af = red_block_sum(a, sm);
  const float bf = red_block_sum(b, sm);
  if (threadIdx.x == 0) {
    atomicAdd(&sums[3 * c], af);
    atomicAdd(&sums[3 * c + 1], bf);
  }
  if (ACT == DPF_ACT_PRELU) {
    const float slf = red_block_sum(sl, sm);
    if (threadIdx.x == 0) atomicAdd(&sums[3 * c + 2], slf);
  }
}

template <int ACT, bool RES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ w, const float* __restrict__ bsh, int wmod,
                                                           const float* __restrict__ res, int act, const float* __restrict__ slope_p,
                                                           float slope_c, const float* __restrict__ sums, float inv_count,
                                                           int training, float* __restrict__ dx, float* __restrict__ dres, int C,
                                                           long long S, const float* __restrict__ count_dev, int gC = 0, int gc0 = 0,
                                                           float* __restrict__ fin_dweight = nullptr, float* __restrict__ fin_dbias = nullptr,
                                                           float* __restrict__ fin_dslope = nullptr) {
  const int row = blockIdx.y;
  // the parameter gradients (bn_bwd_finalize_kernel's job, same sums in the same order) ride on the first workgroup when reduce and
  // apply run in one call: 121 five-microsecond launches per step less
  if (blockIdx.x == 0 && blockIdx.y == 0 && (fin_dweight || fin_dbias || fin_dslope)) {
    for (int cp = threadIdx.x; cp < wmod; cp += 256) {
      float a = 0.f, b = 0.f;
      for (int c = cp; c < C; c += wmod) {
        a += sums[3 * c];
        b += sums[3 * c + 1];
      }
      if (fin_dbias) fin_dbias[cp] = a;
      if (fin_dweight) fin_dweight[cp] = b;
    }
    if (fin_dslope && threadIdx.x == 0) {
      float sl = 0.f;
      for (int c = 0; c < C; ++c) sl += sums[3 * c + 2];
      fin_dslope[0] = sl;
    }
  }
  if (gC > 0) dy += (((long long)(row / (C)) * gC + gc0 + (row % C)) - row) * S;
  if (count_dev) inv_count = 1.f / count_dev[0];     // SyncBatchNorm: the global element count, summed over the ranks on the device
  const int c = row % C;
  float mu = 0.f, is = 1.f, g = 1.f, be = 0.f, m_dz = 0.f, m_dzx = 0.f;
  if (mean) {
    mu = mean[c];
    is = invstd[c];
    g = w ? w[c % wmod] : 1.f;
    be = bsh ? bsh[c % wmod] : 0.f;
    if (training) {
      m_dz = sums[3 * c] * inv_count;
      m_dzx = sums[3 * c + 1] * inv_count;
    }
  }
  const float slope = slope_p ? slope_p[0] : slope_c;
  const float gi = g * is;
  const long long base = (long long)row * S;
  const long long s0 = (long long)blockIdx.x * ROW_CHUNK;
  const long long s1 = min(S, s0 + ROW_CHUNK);
  auto one = [&](float xv, float rv, float d, float& o_dz, float& o_dx) {
    const float xh = (xv - mu) * is;
    const float z = fmaf(xh, g, be) + rv;
    o_dz = d * dact_t<ACT>(z, slope);
    o_dx = mean ? gi * (o_dz - m_dz - xh * m_dzx) : o_dz;
  };
  (void)act;
  if ((S & 3) == 0) {
#pragma unroll 2
    for (long long s = s0 + 4 * threadIdx.x; s < s1; s += 4 * 256) {
      const v4f xv = ld4(x + base + s);                  // last use of x and dy in the backward pass: nontemporal
      const v4f dv = ld4(dy + base + s);
      v4f rv = {0.f, 0.f, 0.f, 0.f};
      if (RES) rv = ld4(res + base + s);
      float z0, z1, z2, z3, x0, x1, x2, x3;
      one(xv.x, rv.x, dv.x, z0, x0); one(xv.y, rv.y, dv.y, z1, x1);
      one(xv.z, rv.z, dv.z, z2, x2); one(xv.w, rv.w, dv.w, z3, x3);
      if (dres) *reinterpret_cast<v4f*>(dres + base + s) = v4f{z0, z1, z2, z3};
      if (dx) *reinterpret_cast<v4f*>(dx + base + s) = v4f{x0, x1, x2, x3};
    }
  } else {
    for (long long s = s0 + threadIdx.x; s < s1; s += 256) {
      float oz, ox;
      one(x[base + s], RES ? res[base + s] : 0.f, dy[base + s], oz, ox);
      if (dres) dres[base + s] = oz;
      if (dx) dx[base + s] = ox;
    }
  }
}

// dweight[c'] = sum_{c % wmod == c'} sums[c][1], dbias likewise with [0], dslope = sum_c sums[c][2]
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums, int C, int wmod, float* __restrict__ dweight,
                                       float* __restrict__ dbias, float* __restrict__ dslope) {
  const int cp = blockIdx.x * blockDim.x + threadIdx.x;
  if (cp < wmod) {
    float a = 0.f, b = 0.f;
    for (int c = cp; c < C; c += wmod) {
      a += sums[3 * c];
      b += sums[3 * c + 1];
    }
    if (dbias) dbias[cp] = a;        // one thread owns channel cp: plain stores, the outputs need no zero-initialisation
    if (dweight) dweight[cp] = b;
  }
  if (dslope && cp == 0) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += sums[3 * c + 2];
    dslope[0] = s;
  }
}

// per-channel sum of g[N,C,S] -> out[C] (+=): conv bias gradients.  DET: one workgroup per channel (see bn_stats_kernel)
template <bool DET>
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ g, float* __restrict__ out, int N, int C, long long S, int chunk) {
  __shared__ float sm[8];
  typedef typename RedAcc<DET>::type acc_t;
  const int c = blockIdx.y % C;
  acc_t a = 0;
  for (int row = blockIdx.y; row < (DET ? N * C : blockIdx.y + 1); row += C) {
    const long long base = (long long)row * S;
    const long long s0 = DET ? 0 : (long long)blockIdx.x * chunk;
    const long long s1 = DET ? S : min(S, s0 + chunk);
    if ((S & 3) == 0) {
#pragma unroll 4
      for (long long s = s0 + 4 * threadIdx.x; s < s1; s += 4 * 256) {
        const float4 v = *reinterpret_cast<const float4*>(g + base + s);
        a += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (long long s = s0 + threadIdx.x; s < s1; s += 256) a += g[base + s];
    }
  }
  const float af = red_block_sum(a, sm);
  if (threadIdx.x == 0) atomicAdd(&out[c], af);
}

inline dim3 row_grid(int rows, long long S) { return dim3((unsigned)dpf_div_up(S, ROW_CHUNK), (unsigned)rows); }

// reduction kernels end in a block reduction + atomics, so they take larger row chunks while the grid still fills the chip
// reduction kernels end in a block reduction + atomics, so they take larger row chunks while the grid still fills the chip (a strided
// variant -- ~8192 blocks looping over 4096-element pieces -- measured the same: tools/debug/bn_sweep.py)
inline int reduce_chunk(int rows, long long S) {
  long long chunk = ((S * rows / 4096 + 1023) / 1024) * 1024;
  if (chunk < ROW_CHUNK) chunk = ROW_CHUNK;
  if (chunk > 65536) chunk = 65536;
  return (int)chunk;
}
inline dim3 reduce_grid(int rows, long long S, int chunk) { return dim3((unsigned)dpf_div_up(S, chunk), (unsigned)rows); }

// deterministic mode: one workgroup per channel (grid (1, C)) instead of one per (row chunk, row)
inline void launch_bn_stats(const float* x, float* ws, int N, int C, long long S, int chunk, hipStream_t st) {
  if (dpf_deterministic()) hipLaunchKernelGGL(bn_stats_kernel<true>, dim3(1, (unsigned)C), dim3(256), 0, st, x, ws, N, C, S, chunk);
  else hipLaunchKernelGGL(bn_stats_kernel<false>, reduce_grid(N * C, S, chunk), dim3(256), 0, st, x, ws, N, C, S, chunk);
}

}  // namespace

extern "C" {

// Training statistics of x[N,C,S].  ws: >= 2*C floats (zeroed here).  Writes mean/invstd [C]; updates the running
// statistics in place when running_mean != NULL (momentum, unbiased variance -- nn.BatchNorm semantics).
int dpf_bn_stats(const float* x, int N, int C, long long S, float eps, float momentum, float* running_mean, float* running_var,
                 float* mean, float* invstd, float* ws, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !mean || !invstd || !ws || N <= 0 || C <= 0 || S <= 0 || (long long)N * C > 65535) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, sizeof(float) * 2 * C, st) != hipSuccess) return DPF_ERR_LAUNCH;
  const int chunk = reduce_chunk(N * C, S);
  launch_bn_stats(x, ws, N, C, S, chunk, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dpf_div_up(C, 64)), dim3(64), 0, st, x, ws, C, S, (double)N * (double)S, eps, momentum,
                     running_mean, running_var, mean, invstd);
  return dpf_check_launch();
}

// Batch statistics from the partial rows of dpf_conv_forward_stats (slab [parts][C][2] doubles, count = N * S elements per channel);
// outputs and running-statistics update as dpf_bn_stats.
int dpf_bn_finalize_partials(double* slab, int parts, int C, long long count, float eps, float momentum, float* running_mean,
                             float* running_var, float* mean, float* invstd, void* stream) {
  dpf_clear_error();
  if (!slab || !mean || !invstd || parts <= 0 || C <= 0 || count <= 0) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  // the folded rows go behind the tile rows: the slab holds >= (parts + FIN_G) rows (dpf_conv_stats_slab_doubles)
  double* part2 = slab + (long long)parts * C * 2;
  const int G = parts < FIN_G ? parts : FIN_G;
  hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(G), dim3(256), 0, st, slab, parts, C, part2);
  hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3(C), dim3(64), 0, st, part2, G, C, (double)count, eps, momentum,
                     running_mean, running_var, mean, invstd);
  return dpf_check_launch();
}

// SyncBatchNorm, step 1 of 2: this rank's per-channel { mean, M2 } of x[N,C,S] -> moments[2*C].  ws: >= 2*C floats.
// The caller all-gathers moments (and N*S) over the ranks and hands the [W][2*C] stack to dpf_bn_merge_moments.
// (torch.nn.SyncBatchNorm semantics, which the reference turns on under DDP: config_manager.py:57, main.py:55.)
int dpf_bn_local_moments(const float* x, int N, int C, long long S, float* moments, float* ws, void* stream) {
  dpf_clear_error();
  if (!x || !moments || !ws || N <= 0 || C <= 0 || S <= 0 || (long long)N * C > 65535) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, sizeof(float) * 2 * C, st) != hipSuccess) return DPF_ERR_LAUNCH;
  const int chunk = reduce_chunk(N * C, S);
  launch_bn_stats(x, ws, N, C, S, chunk, st);
  hipLaunchKernelGGL(bn_local_moments_kernel, dim3(dpf_div_up(C, 64)), dim3(64), 0, st, x, ws, C, S, (double)N * (double)S, moments);
  return dpf_check_launch();
}

// SyncBatchNorm, step 2 of 2: merge W ranks' moments (device, [W][2*C]) and element counts (device, [W]) into the global
// mean / invstd [C]; updates the running statistics with the global unbiased variance when running_mean != NULL.
int dpf_bn_merge_moments(const float* moments, const float* counts, int W, int C, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* invstd, void* stream) {
  dpf_clear_error();
  if (!moments || !counts || !mean || !invstd || W <= 0 || C <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_merge_moments_kernel, dim3(dpf_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream, moments, counts, W, C, eps,
                     momentum, running_mean, running_var, mean, invstd);
  return dpf_check_launch();
}

int dpf_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!running_mean || !running_var || !mean || !invstd || C <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(dpf_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream, running_mean, running_var, C, eps,
                     mean, invstd);
  return dpf_check_launch();
}

// y = act((x-mean)*invstd*w + b + res) + res2 ; mean == NULL -> pure activation/add.  slope: device pointer (PReLU) or NULL
// (then slope_const is used, LeakyReLU).  w/b are indexed c % wmod (instance norm over a [1, B*C, S] view: wmod = C).
int dpf_norm_act_forward(const float* x, const float* mean, const float* invstd, const float* w, const float* b, int wmod,
                         const float* res, const float* res2, int act, const float* slope, float slope_const, float* y, int N, int C,
                         long long S, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !y || N <= 0 || C <= 0 || S <= 0 || (long long)N * C > 65535) return DPF_ERR_INVALID_ARG;
  if (wmod <= 0) wmod = C;
#define DPF_CALL(A, R) hipLaunchKernelGGL((bn_apply_kernel<A, R>), row_grid(N * C, S), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, w, b, wmod, res, res2, act, slope, slope_const, y, C, S, 0, 0)
  DPF_ACT_DISPATCH(act, (res != nullptr || res2 != nullptr), DPF_CALL)
#undef DPF_CALL
  return dpf_check_launch();
}

// dpf_norm_act_forward writing into the channel slice [y_c0, y_c0 + C) of y [N, y_channels, S]: the normalised branches of a
// torch.cat(dim=1) go straight into the concatenated tensor (DPBlock.conv_dilate, src/model/stereodpnet/modules.py:43-45)
int dpf_norm_act_forward_slice(const float* x, const float* mean, const float* invstd, const float* w, const float* b, int wmod,
                               const float* res, const float* res2, int act, const float* slope, float slope_const, float* y, int y_channels,
                               int y_c0, int N, int C, long long S, void* stream) {
  dpf_clear_error();
  if (!x || !y || N <= 0 || C <= 0 || S <= 0 || (long long)N * C > 65535 || y_c0 < 0 || y_c0 + C > y_channels) return DPF_ERR_INVALID_ARG;
  if (wmod <= 0) wmod = C;
#define DPF_CALL(A, R) hipLaunchKernelGGL((bn_apply_kernel<A, R>), row_grid(N * C, S), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, w, b, wmod, res, res2, act, slope, slope_const, y, C, S, y_channels, y_c0)
  DPF_ACT_DISPATCH(act, (res != nullptr || res2 != nullptr), DPF_CALL)
#undef DPF_CALL
  return dpf_check_launch();
}

// Backward of dpf_norm_act_forward w.r.t. x, res, w, b, slope (res2's gradient is dy itself).
// ws: >= 3*C floats.  dweight/dbias [wmod], dslope [1] are WRITTEN (=); any may be NULL.
// phase 0: everything.  phase 1: only the per-channel reductions (ws[3c] = sum dz, ws[3c+1] = sum dz*xhat) and the parameter
// gradients; phase 2: only dx / dres from a ws the caller has summed over the ranks, with count = global N*S (SyncBatchNorm), or
// count < 0: the global count is read from ws[3*C] (the caller put its local count there before the all-reduce: no host sync).
static int norm_act_backward_impl(const float* x, const float* dy, int gC, int gc0, const float* mean, const float* invstd, const float* w,
                                  const float* b, int wmod, const float* res, int act, const float* slope, float slope_const, int training,
                                  float* dx, float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C, long long S,
                                  int phase, double count, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !dy || !ws || N <= 0 || C <= 0 || S <= 0 || (long long)N * C > 65535 || phase < 0 || phase > 3) return DPF_ERR_INVALID_ARG;
  if (wmod <= 0) wmod = C;
  hipStream_t st = (hipStream_t)stream;
  const bool need_reduce = (mean != nullptr) || (act == DPF_ACT_PRELU && dslope);
  bool fused_fin = false;
  if (need_reduce && phase != 2) {
    // phase 3: as phase 0 with a ws the caller guarantees to be zero (a slot of a pre-zeroed arena: one memset per arena, not per layer)
    if (phase != 3 && hipMemsetAsync(ws, 0, sizeof(float) * 3 * C, st) != hipSuccess) return DPF_ERR_LAUNCH;
    const int chunk = reduce_chunk(N * C, S);
    const bool det = dpf_deterministic() != 0;
#define DPF_CALL(A, R)                                                                                                                              \
  do {                                                                                                                                              \
    if (det) hipLaunchKernelGGL((bn_bwd_reduce_kernel<A, R, true>), dim3(1, (unsigned)C), dim3(256), 0, st, x, dy, mean, invstd, w, b, wmod, res, \
                                act, slope, slope_const, ws, N, C, S, chunk, gC, gc0);                                                             \
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<A, R, false>), reduce_grid(N * C, S, chunk), dim3(256), 0, st, x, dy, mean, invstd, w, b, wmod,   \
                            res, act, slope, slope_const, ws, N, C, S, chunk, gC, gc0);                                                             \
  } while (0)
    DPF_ACT_DISPATCH(act, (res != nullptr), DPF_CALL)
#undef DPF_CALL
    fused_fin = (dx || dres) && phase != 1;       // the apply launch below writes the parameter gradients
    if (!fused_fin)
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(dpf_div_up(wmod, 64)), dim3(64), 0, st, ws, C, wmod, mean ? dweight : nullptr,
                         mean ? dbias : nullptr, act == DPF_ACT_PRELU ? dslope : nullptr);
  }
  if ((dx || dres) && phase != 1) {
    // instance-norm view (wmod < C): statistics are per row, count = S; batch norm: count = N*S
    // phase 2 with count < 0: ws[3*C] holds the global element count (the caller all-reduced its local count with the sums)
    const float* count_dev = (phase == 2 && count < 0) ? ws + 3 * (long long)C : nullptr;
    if (count <= 0) count = (double)N * (double)S;
    const float inv_cnt = (float)(1.0 / count);
    float* fdw = (fused_fin && mean) ? dweight : nullptr;
    float* fdb = (fused_fin && mean) ? dbias : nullptr;
    float* fds = (fused_fin && act == DPF_ACT_PRELU) ? dslope : nullptr;
#define DPF_CALL(A, R) hipLaunchKernelGGL((bn_bwd_apply_kernel<A, R>), row_grid(N * C, S), dim3(256), 0, st, x, dy, mean, invstd, w, b, wmod, res, act, slope, slope_const, ws, inv_cnt, training, dx, dres, C, S, count_dev, gC, gc0, fdw, fdb, fds)
    DPF_ACT_DISPATCH(act, (res != nullptr), DPF_CALL)
#undef DPF_CALL
  }
  return dpf_check_launch();
}

int dpf_norm_act_backward_ex(const float* x, const float* dy, const float* mean, const float* invstd, const float* w, const float* b,
                             int wmod, const float* res, int act, const float* slope, float slope_const, int training, float* dx,
                             float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C, long long S, int phase,
                             double count, void* stream) {
  return norm_act_backward_impl(x, dy, 0, 0, mean, invstd, w, b, wmod, res, act, slope, slope_const, training, dx, dres, dweight, dbias,
                                dslope, ws, N, C, S, phase, count, stream);
}

// dpf_norm_act_backward with dy given as the channel slice [dy_c0, dy_c0 + C) of a [N, dy_channels, S] tensor (the gradient of
// the concatenation the forward wrote into)
int dpf_norm_act_backward_slice(const float* x, const float* dy, int dy_channels, int dy_c0, const float* mean, const float* invstd,
                                const float* w, const float* b, int wmod, const float* res, int act, const float* slope, float slope_const,
                                int training, float* dx, float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C,
                                long long S, void* stream) {
  if (dy_c0 < 0 || dy_c0 + C > dy_channels) return DPF_ERR_INVALID_ARG;
  return norm_act_backward_impl(x, dy, dy_channels, dy_c0, mean, invstd, w, b, wmod, res, act, slope, slope_const, training, dx, dres,
                                dweight, dbias, dslope, ws, N, C, S, 0, 0.0, stream);
}

// the same with the phases of dpf_norm_act_backward_ex (SyncBatchNorm of a branch that wrote a channel slice of a concatenation)
int dpf_norm_act_backward_slice_ex(const float* x, const float* dy, int dy_channels, int dy_c0, const float* mean, const float* invstd,
                                   const float* w, const float* b, int wmod, const float* res, int act, const float* slope, float slope_const,
                                   int training, float* dx, float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C,
                                   long long S, int phase, double count, void* stream) {
  if (dy_c0 < 0 || dy_c0 + C > dy_channels) return DPF_ERR_INVALID_ARG;
  return norm_act_backward_impl(x, dy, dy_channels, dy_c0, mean, invstd, w, b, wmod, res, act, slope, slope_const, training, dx, dres,
                                dweight, dbias, dslope, ws, N, C, S, phase, count, stream);
}

int dpf_norm_act_backward(const float* x, const float* dy, const float* mean, const float* invstd, const float* w, const float* b,
                          int wmod, const float* res, int act, const float* slope, float slope_const, int training, float* dx,
                          float* dres, float* dweight, float* dbias, float* dslope, float* ws, int N, int C, long long S, void* stream) {
  return dpf_norm_act_backward_ex(x, dy, mean, invstd, w, b, wmod, res, act, slope, slope_const, training, dx, dres, dweight, dbias, dslope,
                                  ws, N, C, S, 0, 0.0, stream);
}

// out[C] += sum over n, s of g[N,C,S]
int dpf_channel_sum(const float* g, float* out, int N, int C, long long S, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !out || N <= 0 || C <= 0 || S <= 0 || (long long)N * C > 65535) return DPF_ERR_INVALID_ARG;
  const int chunk = reduce_chunk(N * C, S);
  if (dpf_deterministic()) hipLaunchKernelGGL(channel_sum_kernel<true>, dim3(1, (unsigned)C), dim3(256), 0, (hipStream_t)stream, g, out, N, C, S, chunk);
  else hipLaunchKernelGGL(channel_sum_kernel<false>, reduce_grid(N * C, S, chunk), dim3(256), 0, (hipStream_t)stream, g, out, N, C, S, chunk);
  return dpf_check_launch();
}

}  // extern "C"
