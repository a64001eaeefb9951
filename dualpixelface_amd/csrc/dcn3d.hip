// Deformable 3-D convolution (D3D), im2col-free, 64-bit indexing.
//
// Drop-in for the reference's CUDA extension `DCN` (src/module/dcn3d/src/vision.cpp:4-7;
// deform_conv_cuda.cu:18-285; kernels deform_im2col_cuda.cuh:26-405) for group = deformable_group = 1:
//   forward : out[b,k,p] = bias[k] + sum_{c,t} W[k,c,t] * trilinear(x[b,c], base(p,t) + offset[b,3t..3t+2,p])
//   backward: grad_input (adjoint of the sampler), grad_offset (d sample / d coord), grad_weight, grad_bias
// The reference materialises `columns` [C*27, B*P] (2.7 GB per 1024x1536 sample, int32 indices); here a workgroup
// owns 64 output voxels, builds the 27 sampled [C x 64] slices one tap at a time in LDS and contracts them on the fp32
// matrix cores against the [K x C] weight slice of that tap, so columns never reach HBM.
#include "dpf_common.h"
#include "dpf_repack.h"

namespace {

constexpr int TP = 64;          // output voxels per workgroup
constexpr int SP = TP + 1;      // padded LDS row
constexpr int MAXC = 128;

struct DcnP {
  int B, C, K;
  int D, H, W;        // input dims
  int Do, Ho, Wo;     // output dims
  int kd, kh, kw, T;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int CP;             // C rounded up to even
  long long P;        // Do*Ho*Wo
  int tiles_per_b;
  int nchunk;
};

struct Corner {   // per output voxel and tap
  int d0, h0, w0;
  float ld, lh, lw;
  int valid;
};

__device__ __forceinline__ Corner make_corner(const DcnP& p, const float* __restrict__ off_b, int t, long long pos) {
  Corner c;
  c.valid = 0;
  c.d0 = c.h0 = c.w0 = 0;
  c.ld = c.lh = c.lw = 0.f;
  if (pos >= p.P) return c;
  const int xo = (int)(pos % p.Wo);
  const int yo = (int)((pos / p.Wo) % p.Ho);
  const int zo = (int)(pos / ((long long)p.Wo * p.Ho));
  const int tk = t % p.kw, tj = (t / p.kw) % p.kh, ti = t / (p.kw * p.kh);
  const float od = off_b[(long long)(3 * t) * p.P + pos];
  const float oh = off_b[(long long)(3 * t + 1) * p.P + pos];
  const float ow = off_b[(long long)(3 * t + 2) * p.P + pos];
  const float fd = (float)(zo * p.sd - p.pd + ti * p.dd) + od;
  const float fh = (float)(yo * p.sh - p.ph + tj * p.dh) + oh;
  const float fw = (float)(xo * p.sw - p.pw + tk * p.dw) + ow;
  if (fd > -1.f && fh > -1.f && fw > -1.f && fd < (float)p.D && fh < (float)p.H && fw < (float)p.W) {   // cuh:248
    const float d0 = floorf(fd), h0 = floorf(fh), w0 = floorf(fw);
    c.d0 = (int)d0; c.h0 = (int)h0; c.w0 = (int)w0;
    c.ld = fd - d0; c.lh = fh - h0; c.lw = fw - w0;
    c.valid = 1;
  }
  return c;
}

// corner j = (jd, jh, jw) bits; returns flat voxel index or -1 (cuh:43-65), weight (cuh:67-68)
__device__ __forceinline__ long long corner_index(const DcnP& p, const Corner& c, int j, float& wgt) {
  const int jd = (j >> 2) & 1, jh = (j >> 1) & 1, jw = j & 1;
  const int d = c.d0 + jd, h = c.h0 + jh, w = c.w0 + jw;
  wgt = (jd ? c.ld : 1.f - c.ld) * (jh ? c.lh : 1.f - c.lh) * (jw ? c.lw : 1.f - c.lw);
  if (!c.valid || d < 0 || d > p.D - 1 || h < 0 || h > p.H - 1 || w < 0 || w > p.W - 1) return -1;
  return ((long long)d * p.H + h) * p.W + w;
}

// S[c][pp] = trilinear sample of channel c at voxel pp of the tile, for tap t
__device__ __forceinline__ void build_samples(const DcnP& p, const float* __restrict__ xb, const Corner& cn, float* s_S, int tid) {
  const int pp = tid & 63, q = tid >> 6;
  long long idx[8];
  float wg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) idx[j] = corner_index(p, cn, j, wg[j]);
  const long long chan = (long long)p.D * p.H * p.W;
  for (int c = q; c < p.CP; c += 4) {
    float v = 0.f;
    if (c < p.C) {
      const float* xc = xb + (long long)c * chan;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (idx[j] >= 0) v += wg[j] * xc[idx[j]];
    }
    s_S[c * SP + pp] = v;
  }
}

// ------------------------------------------------------------------------------------------ forward
template <int MT>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                      const float* __restrict__ wt /*[T][C][KT]*/, const float* __restrict__ bias,
                                                      float* __restrict__ out, DcnP p) {
  extern __shared__ __align__(16) float smem[];
  float* s_S = smem;   // [CP][SP]
  constexpr int KT = 32 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x / p.tiles_per_b;
  const long long pos0 = (long long)(blockIdx.x % p.tiles_per_b) * TP;
  const float* xb = x + (long long)b * p.C * p.D * p.H * p.W;
  const float* off_b = offset + (long long)b * 3 * p.T * p.P;

  constexpr int NTILES = MT * 2;                 // (m, nt) tiles of 32x32
  constexpr int TPW = (NTILES + 3) / 4;          // tiles per wave
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  for (int t = 0; t < p.T; ++t) {
    const Corner cn = make_corner(p, off_b, t, pos0 + (tid & 63));
    __syncthreads();
    build_samples(p, xb, cn, s_S, tid);
    __syncthreads();
    const float* wtt = wt + (long long)t * p.C * KT;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tile = wave + 4 * i;
      if (tile < NTILES) {
        const int m = tile >> 1, nt = tile & 1;
        for (int cp = 0; cp < p.CP / 2; ++cp) {
          const int c = 2 * cp + hh;
          const float a = c < p.C ? wtt[(long long)c * KT + m * 32 + l31] : 0.f;
          const float bv = s_S[c * SP + nt * 32 + l31];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[i], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tile = wave + 4 * i;
    if (tile < NTILES) {
      const int m = tile >> 1, nt = tile & 1;
      const long long pos = pos0 + nt * 32 + l31;
      if (pos < p.P) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (k < p.K) out[((long long)b * p.K + k) * p.P + pos] = acc[i][j] + (bias ? bias[k] : 0.f);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: input + offset
// MTC = ceil(CP/32) row tiles of gcol[c][p] = sum_k W[k][c][t] * go[k][p]
template <int MTC>
__global__ __launch_bounds__(256) void dcn_bwd_data_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                           const float* __restrict__ wt2 /*[T][K][CT]*/, const float* __restrict__ go,
                                                           float* __restrict__ dx, float* __restrict__ doff, DcnP p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int CT = 32 * MTC;
  float* s_go = smem;                    // [K][SP]
  float* s_gc = s_go + p.K * SP;         // [CT][SP]
  float* s_red = s_gc + CT * SP;         // [3][4][TP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x / p.tiles_per_b;
  const long long pos0 = (long long)(blockIdx.x % p.tiles_per_b) * TP;
  const long long chan = (long long)p.D * p.H * p.W;
  const float* xb = x + (long long)b * p.C * chan;
  float* dxb = dx + (long long)b * p.C * chan;
  const float* off_b = offset + (long long)b * 3 * p.T * p.P;
  float* doff_b = doff + (long long)b * 3 * p.T * p.P;

  for (int i = tid; i < p.K * TP; i += 256) {
    const int k = i / TP, pp = i - k * TP;
    const long long pos = pos0 + pp;
    s_go[k * SP + pp] = pos < p.P ? go[((long long)b * p.K + k) * p.P + pos] : 0.f;
  }
  constexpr int NTILES = MTC * 2;
  constexpr int TPW = (NTILES + 3) / 4;
  const int pp = tid & 63, q = tid >> 6;

  for (int t = 0; t < p.T; ++t) {
    const Corner cn = make_corner(p, off_b, t, pos0 + pp);
    __syncthreads();   // s_go ready / previous tap's s_gc, s_red consumed
    const float* wtt = wt2 + (long long)t * p.K * CT;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tile = wave + 4 * i;
      if (tile < NTILES) {
        const int m = tile >> 1, nt = tile & 1;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int kp = 0; kp < (p.K + 1) / 2; ++kp) {
          const int k = 2 * kp + hh;
          const float a = k < p.K ? wtt[(long long)k * CT + m * 32 + l31] : 0.f;
          const float bv = k < p.K ? s_go[k * SP + nt * 32 + l31] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int c = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
          s_gc[c * SP + nt * 32 + l31] = acc[j];
        }
      }
    }
    __syncthreads();
    // scatter + coordinate gradients: thread = (voxel pp, channel residue q)
    float gd = 0.f, gh = 0.f, gw = 0.f;
    if (cn.valid) {
      long long idx[8];
      float wg[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) idx[j] = corner_index(p, cn, j, wg[j]);
      for (int c = q; c < p.C; c += 4) {
        const float gcv = s_gc[c * SP + pp];
        const float* xc = xb + (long long)c * chan;
        float* dxc = dxb + (long long)c * chan;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (idx[j] < 0) continue;
          const int jd = (j >> 2) & 1, jh = (j >> 1) & 1, jw = j & 1;
          atomicAdd(&dxc[idx[j]], wg[j] * gcv);                                  // cuh:313-331
          const float v = xc[idx[j]] * gcv;
          const float fd = jd ? cn.ld : 1.f - cn.ld, fh = jh ? cn.lh : 1.f - cn.lh, fw = jw ? cn.lw : 1.f - cn.lw;
          gd += (jd ? 1.f : -1.f) * fh * fw * v;                                  // cuh:131-187
          gh += (jh ? 1.f : -1.f) * fd * fw * v;
          gw += (jw ? 1.f : -1.f) * fd * fh * v;
        }
      }
    }
    s_red[(0 * 4 + q) * TP + pp] = gd;
    s_red[(1 * 4 + q) * TP + pp] = gh;
    s_red[(2 * 4 + q) * TP + pp] = gw;
    __syncthreads();
    if (tid < 3 * TP) {
      const int dir = tid / TP, p2 = tid - dir * TP;
      const long long pos = pos0 + p2;
      if (pos < p.P) {
        const float* r = s_red + dir * 4 * TP + p2;
        doff_b[(long long)(3 * t + dir) * p.P + pos] = (r[0] + r[TP]) + (r[2 * TP] + r[3 * TP]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: weight
// grid = T * nchunk; block = one tap, a strided set of voxel tiles; dW[k][c][t] += sum_p go[k][p] * S[c][p]
template <int MT, int MTC>
__global__ __launch_bounds__(256) void dcn_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                        const float* __restrict__ go, float* __restrict__ dw, DcnP p) {
  extern __shared__ __align__(16) float smem[];
  float* s_S = smem;                 // [32*MTC][SP]
  float* s_go = s_S + 32 * MTC * SP; // [32*MT][SP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int t = blockIdx.x / p.nchunk;
  const int chunk = blockIdx.x % p.nchunk;
  constexpr int NTILES = MT * MTC;
  constexpr int TPW = (NTILES + 3) / 4;
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  // zero the padded rows once
  for (int i = tid; i < 32 * MTC * SP; i += 256) s_S[i] = 0.f;
  const long long ntile = (long long)p.B * p.tiles_per_b;
  const long long chan = (long long)p.D * p.H * p.W;
  for (long long tile = chunk; tile < ntile; tile += p.nchunk) {
    const int b = (int)(tile / p.tiles_per_b);
    const long long pos0 = (tile % p.tiles_per_b) * TP;
    const float* xb = x + (long long)b * p.C * chan;
    const float* off_b = offset + (long long)b * 3 * p.T * p.P;
    const Corner cn = make_corner(p, off_b, t, pos0 + (tid & 63));
    __syncthreads();
    build_samples(p, xb, cn, s_S, tid);
    for (int i = tid; i < 32 * MT * TP; i += 256) {
      const int k = i / TP, pp = i - k * TP;
      const long long pos = pos0 + pp;
      s_go[k * SP + pp] = (k < p.K && pos < p.P) ? go[((long long)b * p.K + k) * p.P + pos] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tl = wave + 4 * i;
      if (tl < NTILES) {
        const int m = tl / MTC, mc = tl - m * MTC;
#pragma unroll 4
        for (int ps = 0; ps < TP / 2; ++ps) {
          const int pp = 2 * ps + hh;
          const float a = s_go[(m * 32 + l31) * SP + pp];
          const float bv = s_S[(mc * 32 + l31) * SP + pp];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[i], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tl = wave + 4 * i;
    if (tl < NTILES) {
      const int m = tl / MTC, mc = tl - m * MTC;
      const int c = mc * 32 + l31;
      if (c < p.C) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (k < p.K) atomicAdd(&dw[((long long)k * p.C + c) * p.T + t], acc[i][j]);
        }
      }
    }
  }
}

int fill_params(DcnP& p, int B, int C, int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                int dd, int dh, int dw) {
  if (B <= 0 || C <= 0 || K <= 0 || C > MAXC || K > MAXC) return DPF_ERR_UNSUPPORTED;
  p.B = B; p.C = C; p.K = K; p.D = D; p.H = H; p.W = W;
  p.kd = kd; p.kh = kh; p.kw = kw; p.T = kd * kh * kw;
  p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.Do = (D + 2 * pd - (dd * (kd - 1) + 1)) / sd + 1;
  p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (p.Do <= 0 || p.Ho <= 0 || p.Wo <= 0 || p.T > 64) return DPF_ERR_INVALID_ARG;
  p.P = (long long)p.Do * p.Ho * p.Wo;
  p.CP = (C + 1) & ~1;
  p.tiles_per_b = (int)((p.P + TP - 1) / TP);
  p.nchunk = 1;
  return DPF_OK;
}

template <typename F>
int set_lds(F f, size_t lds) {
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return DPF_ERR_LAUNCH;
  return DPF_OK;
}

}  // namespace

extern "C" {

int dpf_channel_sum(const float* g, float* out, int N, int C, long long S, void* stream);   // norm_act.hip

// workspace floats for dpf_deform_conv3d_forward / _backward (repacked weights)
long long dpf_deform_conv3d_workspace_floats(int C, int K, int T) {
  const long long a = (long long)T * C * (((K + 31) / 32) * 32);
  const long long b = (long long)T * K * (((C + 31) / 32) * 32);
  return a > b ? a : b;
}

// Mirrors DCN.deform_conv_forward(input, weight, bias, offset, kd,kh,kw, sd,sh,sw, pd,ph,pw, dd,dh,dw, group, deformable_group,
// im2col_step) (deform_conv.h:10-29); group/deformable_group must be 1; im2col_step is accepted and ignored (no columns).
int dpf_deform_conv3d_forward(const float* input, const float* weight, const float* bias, const float* offset, float* output, float* ws,
                              int B, int C, int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph,
                              int pw, int dd, int dh, int dw, int group, int deformable_group, int im2col_step, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  (void)im2col_step;
  if (!input || !weight || !offset || !output || !ws) return DPF_ERR_INVALID_ARG;
  if (group != 1 || deformable_group != 1) return DPF_ERR_UNSUPPORTED;
  DcnP p{};
  int rc = fill_params(p, B, C, D, H, W, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw);
  if (rc != DPF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int MT = (K + 31) / 32, KT = 32 * MT;
  hipLaunchKernelGGL(repack_weights_kernel, dim3(dpf_ew_grid((long long)p.T * C * KT)), dim3(256), 0, st, weight, ws, K, C, p.T, KT, 0);
  const size_t lds = sizeof(float) * (size_t)p.CP * SP;
  const dim3 grid((unsigned)(B * p.tiles_per_b));
#define DPF_F(M)                                                                                   \
  {                                                                                                \
    if (set_lds(dcn_fwd_kernel<M>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                          \
    hipLaunchKernelGGL((dcn_fwd_kernel<M>), grid, dim3(256), lds, st, input, offset, ws, bias, output, p); \
  }
  switch (MT) { case 1: DPF_F(1); break; case 2: DPF_F(2); break; case 3: DPF_F(3); break; default: DPF_F(4); break; }
#undef DPF_F
  return dpf_check_launch();
}

// Mirrors DCN.deform_conv_backward(...) -> [grad_input, grad_offset, grad_weight, grad_bias] (deform_conv.h:49-69).
// grad_input / grad_weight / grad_bias are zero-initialised here, like the reference's at::zeros_like (cu:202-205).
int dpf_deform_conv3d_backward(const float* input, const float* weight, const float* bias, const float* offset, const float* grad_output,
                               float* grad_input, float* grad_offset, float* grad_weight, float* grad_bias, float* ws, int B, int C,
                               int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
                               int dh, int dw, int group, int deformable_group, int im2col_step, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  (void)im2col_step; (void)bias;
  if (!input || !weight || !offset || !grad_output || !grad_input || !grad_offset || !grad_weight || !ws) return DPF_ERR_INVALID_ARG;
  if (group != 1 || deformable_group != 1) return DPF_ERR_UNSUPPORTED;
  DcnP p{};
  int rc = fill_params(p, B, C, D, H, W, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw);
  if (rc != DPF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int MT = (K + 31) / 32, MTC = (p.CP + 31) / 32, CT = 32 * MTC;
  const long long in_elems = (long long)B * C * D * H * W;
  if (hipMemsetAsync(grad_input, 0, sizeof(float) * in_elems, st) != hipSuccess) return DPF_ERR_LAUNCH;
  if (hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)K * C * p.T, st) != hipSuccess) return DPF_ERR_LAUNCH;
  // wt2[T][K][CT]: reduce = K (A), out = C (B)
  hipLaunchKernelGGL(repack_weights_kernel, dim3(dpf_ew_grid((long long)p.T * K * CT)), dim3(256), 0, st, weight, ws, K, C, p.T, CT, 1);
  {
    const size_t lds = sizeof(float) * ((size_t)K * SP + (size_t)CT * SP + 3 * 4 * TP);
    const dim3 grid((unsigned)(B * p.tiles_per_b));
#define DPF_D(M)                                                                                                       \
  {                                                                                                                    \
    if (set_lds(dcn_bwd_data_kernel<M>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                                         \
    hipLaunchKernelGGL((dcn_bwd_data_kernel<M>), grid, dim3(256), lds, st, input, offset, ws, grad_output, grad_input, grad_offset, p); \
  }
    switch (MTC) { case 1: DPF_D(1); break; case 2: DPF_D(2); break; case 3: DPF_D(3); break; default: DPF_D(4); break; }
#undef DPF_D
  }
  {
    const long long ntile = (long long)B * p.tiles_per_b;
    long long nchunk = 2048 / p.T;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > ntile) nchunk = ntile;
    p.nchunk = (int)nchunk;
    const size_t lds = sizeof(float) * ((size_t)32 * MTC * SP + (size_t)32 * MT * SP);
    const dim3 grid((unsigned)(p.T * p.nchunk));
#define DPF_W(M, N)                                                                                           \
  {                                                                                                           \
    if (set_lds(dcn_wgrad_kernel<M, N>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                                \
    hipLaunchKernelGGL((dcn_wgrad_kernel<M, N>), grid, dim3(256), lds, st, input, offset, grad_output, grad_weight, p); \
  }
    if (MT <= 2 && MTC <= 2) {
      if (MT == 1 && MTC == 1) DPF_W(1, 1) else if (MT == 1) DPF_W(1, 2) else if (MTC == 1) DPF_W(2, 1) else DPF_W(2, 2)
    } else {
      DPF_W(4, 4)
    }
#undef DPF_W
  }
  if (grad_bias) {
    // grad_bias[k] = sum_{b,p} go[b,k,p]  (cu:277) -- small row reduction
    if (hipMemsetAsync(grad_bias, 0, sizeof(float) * K, st) != hipSuccess) return DPF_ERR_LAUNCH;
    rc = dpf_channel_sum(grad_output, grad_bias, B, K, p.P, stream);
    if (rc != DPF_OK) return rc;
  }
  return dpf_check_launch();
}

}  // extern "C"
