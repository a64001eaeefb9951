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
#include "dcn_internal.h"
#include "conv_internal.h"
#include <cstdlib>

namespace {

// In-kernel s_memtime stamps of one workgroup (-DDPF_STAMPS builds only; diagnostic)
#ifdef DPF_STAMPS
__device__ unsigned long long g_stamps[16 * 128 * 2];
#define DPF_STAMP(step, slot)                                                                                          \
  if (blockIdx.x == 3000 && lane == 0 && (step) < 128) g_stamps[(DPF_STAMP_WAVE * 128 + (step)) * 2 + (slot)] = __builtin_readcyclecounter();
#else
#define DPF_STAMP(step, slot)
#endif

// XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (blockIdx % 8 labels the XCD and its L2), so with
// the plain blockIdx -> tile map x-neighbouring tiles -- which share halo rows, offset / grad_output cache lines -- sit behind eight
// different L2s and every line is fetched from HBM several times.  Here each label walks a CONTIGUOUS range of tiles (x fastest):
// a bijection of [0, n) for any n.
__device__ __forceinline__ int dpf_xcd_tile(int blk, int n) {
  const int q = n >> 3, r = n & 7, x = blk & 7, i = blk >> 3;
  return x * q + (x < r ? x : r) + i;
}

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

struct Off3 {
  float d, h, w;
};

// the three offset components of tap t at output voxel pos (cuh:238-243); zeros beyond the volume / tap range
__device__ __forceinline__ Off3 load_off(const DcnP& p, const float* __restrict__ off_b, int t, long long pos) {
  Off3 o = {0.f, 0.f, 0.f};
  if (pos < p.P && t < p.T) {
    o.d = off_b[(long long)(3 * t) * p.P + pos];
    o.h = off_b[(long long)(3 * t + 1) * p.P + pos];
    o.w = off_b[(long long)(3 * t + 2) * p.P + pos];
  }
  return o;
}

__device__ __forceinline__ Corner corner_from(const DcnP& p, int t, long long pos, const Off3& o) {
  Corner c;
  c.valid = 0;
  c.d0 = c.h0 = c.w0 = 0;
  c.ld = c.lh = c.lw = 0.f;
  if (pos >= p.P) return c;
  const int xo = (int)(pos % p.Wo);
  const int yo = (int)((pos / p.Wo) % p.Ho);
  const int zo = (int)(pos / ((long long)p.Wo * p.Ho));
  const int tk = t % p.kw, tj = (t / p.kw) % p.kh, ti = t / (p.kw * p.kh);
  const float fd = (float)(zo * p.sd - p.pd + ti * p.dd) + o.d;
  const float fh = (float)(yo * p.sh - p.ph + tj * p.dh) + o.h;
  const float fw = (float)(xo * p.sw - p.pw + tk * p.dw) + o.w;
  if (fd > -1.f && fh > -1.f && fw > -1.f && fd < (float)p.D && fh < (float)p.H && fw < (float)p.W) {   // cuh:248
    const float d0 = floorf(fd), h0 = floorf(fh), w0 = floorf(fw);
    c.d0 = (int)d0; c.h0 = (int)h0; c.w0 = (int)w0;
    c.ld = fd - d0; c.lh = fh - h0; c.lw = fw - w0;
    c.valid = 1;
  }
  return c;
}

__device__ __forceinline__ Corner make_corner(const DcnP& p, const float* __restrict__ off_b, int t, long long pos) {
  return corner_from(p, t, pos, load_off(p, off_b, t, pos));
}

// ---- division-free variants for the region kernels: the thread knows its output voxel (zo, yo, xo), the tap loop keeps the
// tap's (ti, tj, tk) in scalars, and the offsets are read through a pointer that advances by 3*P per tap
struct TapIt {
  int ti, tj, tk;
};

__device__ __forceinline__ void tap_next(const DcnP& p, TapIt& it) {
  if (++it.tk == p.kw) {
    it.tk = 0;
    if (++it.tj == p.kh) {
      it.tj = 0;
      ++it.ti;
    }
  }
}

__device__ __forceinline__ Off3 load_off_ptr(const float* __restrict__ offp, long long P, bool ok) {
  Off3 o = {0.f, 0.f, 0.f};
  if (ok) {
    o.d = offp[0];
    o.h = offp[P];
    o.w = offp[2 * P];
  }
  return o;
}

// zb / yb / xb = output voxel * stride - pad
__device__ __forceinline__ Corner corner_at(const DcnP& p, bool pvalid, int zb, int yb, int xb, const TapIt& it, const Off3& o) {
  Corner c;
  c.valid = 0;
  c.d0 = c.h0 = c.w0 = 0;
  c.ld = c.lh = c.lw = 0.f;
  const float fd = (float)(zb + it.ti * p.dd) + o.d;
  const float fh = (float)(yb + it.tj * p.dh) + o.h;
  const float fw = (float)(xb + it.tk * p.dw) + o.w;
  if (pvalid && fd > -1.f && fh > -1.f && fw > -1.f && fd < (float)p.D && fh < (float)p.H && fw < (float)p.W) {   // cuh:248
    const float d0 = floorf(fd), h0 = floorf(fh), w0 = floorf(fw);
    c.d0 = (int)d0; c.h0 = (int)h0; c.w0 = (int)w0;
    c.ld = fd - d0; c.lh = fh - h0; c.lw = fw - w0;
    c.valid = 1;
  }
  return c;
}

// w[A][B][T] -> wt[T][RP][KT] like repack_weights_kernel, with the reduce index zero-padded to RP rows as well, so that the
// kernels can fetch their weight fragments without bounds checks
__global__ void repack_weights_pad_kernel(const float* __restrict__ w, float* __restrict__ wt, int A, int B, int T, int KT, int mode, int RP) {
  const int R = mode == 0 ? B : A;
  const int O = mode == 0 ? A : B;
  const long long total = (long long)T * RP * KT;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i % KT);
    const int r = (int)((i / KT) % RP);
    const int t = (int)(i / ((long long)KT * RP));
    float v = 0.f;
    if (o < O && r < R) {
      const int a = mode == 0 ? o : r;
      const int b = mode == 0 ? r : o;
      v = w[((long long)a * B + b) * T + t];
    }
    wt[i] = v;
  }
}

// corner j = (jd, jh, jw) bits; returns flat voxel index or -1 (cuh:43-65), weight (cuh:67-68)
__device__ __forceinline__ long long corner_index(const DcnP& p, const Corner& c, int j, float& wgt) {
  const int jd = (j >> 2) & 1, jh = (j >> 1) & 1, jw = j & 1;
  const int d = c.d0 + jd, h = c.h0 + jh, w = c.w0 + jw;
  wgt = (jd ? c.ld : 1.f - c.ld) * (jh ? c.lh : 1.f - c.lh) * (jw ? c.lw : 1.f - c.lw);
  if (!c.valid || d < 0 || d > p.D - 1 || h < 0 || h > p.H - 1 || w < 0 || w > p.W - 1) return -1;
  return ((long long)d * p.H + h) * p.W + w;
}

// 32-bit variant (the host checks D*H*W < 2^31)
__device__ __forceinline__ int corner_index32(const DcnP& p, const Corner& c, int j, float& wgt) {
  const int jd = (j >> 2) & 1, jh = (j >> 1) & 1, jw = j & 1;
  const int d = c.d0 + jd, h = c.h0 + jh, w = c.w0 + jw;
  wgt = (jd ? c.ld : 1.f - c.ld) * (jh ? c.lh : 1.f - c.lh) * (jw ? c.lw : 1.f - c.lw);
  if (!c.valid || d < 0 || d > p.D - 1 || h < 0 || h > p.H - 1 || w < 0 || w > p.W - 1) return -1;
  return (d * p.H + h) * p.W + w;
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
template <int MTC, bool DO_DX>
__global__ __launch_bounds__(256) void dcn_bwd_data_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                           const float* __restrict__ wt2 /*[T][K][CT]*/, const float* __restrict__ go,
                                                           float* __restrict__ dx, float* __restrict__ doff, DcnP p, long long* gi_shadow) {
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
          if (DO_DX) dcn_acc_add(dx, gi_shadow, &dxc[idx[j]], wg[j] * gcv);       // cuh:313-331 (fallback path)
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

// ------------------------------------------------------------------------------------------ backward: input, LDS-privatised
// grad_input[b,c,v] = sum_{p,t,corner: voxel(p,t,corner)=v} w * gcol[c,t,p],  gcol = W^T . grad_out.
// The reference scatters every contribution with a global atomicAdd (deform_im2col_cuda.cuh:313-331): 27*8*C atomics per
// output voxel, which on MI355X execute at the memory side.  Here a workgroup owns TZ x 2 x 32 output voxels, keeps the
// haloed input region [RZ][RY][RX] x 16 channels in LDS, computes gcol with v_mfma_f32_16x16x4_f32 so that a lane owns one
// channel of 4 voxels (D row = voxel, col = channel), adds the 8 corner contributions with conflict-free LDS atomics
// (16 consecutive channels per voxel) and flushes the region once with row-contiguous global atomics.  Samples that leave
// the region (|offset| >~ 2) fall back to a direct global atomic.
// grad_input tile: 4 x 8 x 8 output voxels.  With a halo of 3 its LDS region is 4 x 14 x 14 cells = 3.1 per output voxel; the 4 x 2 x 32 tile
// needs 4 x 10 x 40 = 6.25 (twice the region fills, flushes and global float atomics): 8.48 -> 7.97 ms per
// 64-channel launch (4 x 4 x 16: 8.01).  -DDPF_GI_TX=32|16|8 selects the shape at compile time.
#ifndef DPF_GI_TX
#define DPF_GI_TX 8
#endif
// halo of the grad_input region: 4 (the compact tile leaves the LDS for it): at the bench model's offsets (p99 3.7 voxels in the first
// layer) fewer corners take the global far path -- backward 30.7 -> 30.1 ms per step (halo 5: 29.9, with 1.6x the region cells to flush)
#ifndef DPF_GI_R
#define DPF_GI_R 4
#endif
constexpr int GI_TX = DPF_GI_TX, GI_TY = 64 / GI_TX, GI_R = DPF_GI_R, GI_CH = 8;   // tile 4 x GI_TY x GI_TX outputs (64 per plane), halo 3; GI_CH: granularity of the per-chunk max |W| table
constexpr int GI_TXS = GI_TX == 32 ? 5 : (GI_TX == 16 ? 4 : 3);

struct GiP {
  int TZ, RZmax, RY, RX;     // tile depth, region dims
  int tilesZ, tilesY, tilesX;
  int CG;                    // grad_input is produced for channels [0, CG) only
};

// Measured on MI355X (tools/lds_atomic_bench.hip): ds_add_f32 sustains 0.33 lanes/clk/CU whatever the address pattern, ds_add_f64 3.1
// and ds_add_u64 4.8-5.4: the region accumulates in 64-bit FIXED POINT (dcn_bwd_input_pk_kernel below; rounds 1-2 used fp64, then one
// 64-bit fixed-point value per channel -- 8 channels per pass, 17.9 ms per launch where the packed-pair kernel takes 10.0).

// wmax[ch] = max over taps, k and the GI_CH channels of chunk ch of |wt2[t][k][c]|   (one workgroup per chunk)
__global__ __launch_bounds__(256) void dcn_wmax_kernel(const float* __restrict__ wt2, float* __restrict__ wmax, int T, int CT, int C) {
  __shared__ float sm[4];
  const int c0 = blockIdx.x * GI_CH;
  float m = 0.f;
  for (int i = threadIdx.x; i < T * 64 * GI_CH; i += 256) {
    const int cc = i % GI_CH, row = i / GI_CH;
    if (c0 + cc < C) m = fmaxf(m, fabsf(wt2[(long long)row * CT + c0 + cc]));
  }
  m = dpf_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) wmax[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// the same table from the caller's weight tensor w[K][C][T] (the f16-component path repacks straight into fragments: no wt2)
__global__ __launch_bounds__(256) void dcn_wmax_w_kernel(const float* __restrict__ w, float* __restrict__ wmax, int K, int C, int T) {
  __shared__ float sm[4];
  const int c0 = blockIdx.x * GI_CH;
  float m = 0.f;
  for (int i = threadIdx.x; i < K * GI_CH * T; i += 256) {
    const int t = i % T, cc = (i / T) % GI_CH, k = i / (T * GI_CH);
    if (c0 + cc < C) m = fmaxf(m, fabsf(w[((long long)k * C + c0 + cc) * T + t]));
  }
  m = dpf_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) wmax[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// f16 components of the gcol B operand (dcn_bwd_input_pk_kernel<.., F16 = true>, v_mfma_f32_16x16x32_f16):
// wph[tap][chunk of 16 channels][k half mf][hi | lo][lane][8 halves], value i of lane (l15, lg) = component of
// W[k = 32 mf + 8 lg + i][c = 16 chunk + l15][tap] * 2^(141 - E) (zero beyond K / C), E = the largest exponent of the tensor -> *wexp
__global__ void dcn_repack_pk_h_kernel(const float* __restrict__ w, unsigned short* __restrict__ wph, int K, int C, int T, int nch16, int* __restrict__ wexp) {
  __shared__ int s_e[16];
  float m = 0.f;
  const int nw = K * C * T;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
  const int e = dpf_wave_max_exp(__builtin_bit_cast(unsigned, m));
  if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6] = e;
  __syncthreads();
  int E = DPF_H3_EMIN;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) E = s_e[i] > E ? s_e[i] : E;
  E = E > 254 ? 254 : E;
  const float sc = dpf_h3_scale(E);
  if (blockIdx.x == 0 && threadIdx.x == 0) wexp[0] = E;
  const int total = T * nch16 * 2048;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int idx = i & 7, lane = (i >> 3) & 63, comp = (i >> 9) & 1, mf = (i >> 10) & 1;
    const int r = i >> 11;
    const int chunk = r % nch16, t = r / nch16;
    const int k = 32 * mf + 8 * (lane >> 4) + idx, c = 16 * chunk + (lane & 15);
    const float v = (k < K && c < C) ? w[((long long)k * C + c) * T + t] * sc : 0.f;
    unsigned h, l;
    dpf_split_pair_h(v, 0.f, h, l);
    wph[i] = (unsigned short)((comp ? l : h) & 0xffffu);
  }
}

// ------------------------------------------------------------------------------------------ backward: input, packed pairs
// Same LDS-privatised scatter as dcn_bwd_input_kernel<.., FX = true>, with TWO channels per ds_add_u64: a contribution of the
// channel pair (2m, 2m+1) is the signed 64-bit integer  q_even * 2^32 + q_odd  (q = round(w * gcol * qscale), |sum q| < 2^31).
// 64-bit adds are linear, so the cell holds  (sum q_even) * 2^32 + (sum q_odd)  exactly; the flush decodes the low field by sign
// extension and the high field as (total - low) >> 32.  A cell of 8 u64 therefore carries 16 channels: half the passes (table
// phases, region fills, weight-fragment loads) of the 8-channel version, half the atomics per channel (the LDS atomic unit retires
// ~5.4 u64 lanes/clk/CU whatever they carry), and the gcol MFMA chain has 16 distinct columns (the 8-channel kernel mirrors 8).
// Lane l15 of a 16-lane group owns column c0 + l15 of gcol; one DPP quad_perm hands it its neighbour's column, so lanes (2m, 2m+1)
// both hold the pair m and split the 8 corners (even lane: corners 0-3, odd lane: 4-7).
// 32-bit fields need a tight bound on sum |w * gcol| per cell: |gcol| <= gbound * max|W| as before, and the WEIGHT MASS a cell can
// collect (sum of trilinear weights landing on it: ~30-60 for sigma ~ 1 offsets, 6912 if every sample of the tile hit one cell) is
// measured instead of assumed: the first pass scatters the weights themselves into a u32 side region (s_mass) from the table phase
// and runs with the bound MASS0; if the measured mass exceeds it (pathological offset fields) that pass is repeated with the
// measured bound -- far-corner global atomics are emitted on the first attempt only.  Later passes use the measured bound.
// Error per contribution <= 0.5 unit, unit = gbound * wmax * mass / 2^30: ~1e-6 of the tensor scale at MASS0, deterministic
// (integer adds commute).
// floor(x + 0.5) in ONE instruction (__float2int_rn is v_rndne_f32 + v_cvt_i32_f32): the quantiser of the packed scatter runs 64 times per
// (wave, tap); ties round up instead of to even -- the same 0.5-unit bound
__device__ __forceinline__ int cvt_rpi(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

constexpr int PK_CH = 16, PK_CS = 8;
constexpr float PK_MASS0 = 128.f, PK_MASS_Q = 131072.f;   // first-pass mass bound; fixed-point scale of the mass counters (6912 * 2^17 < 2^30)

// F16 = true (default unless dpf_set_f32_matrix_path(0) / DPF_DCN_GCOL16=0): the gcol chain runs on v_mfma_f32_16x16x32_f16 from two f16
// components per operand (conv_internal.h) -- 6 MFMAs per (16 voxels, 16 channels, 64 k) instead of 16 fp32 ones; the fragments keep the
// size of the fp32 ones (the bf16 three-way variant of round 5 spilled: 24 + 24 registers); wt2 then holds dcn_repack_pk_h_kernel's
// fragments and its exponent comes through wexp.  RANGE GUARD (conv_internal.h): gcol[voxel][channel] sums over the output channels of
// ONE voxel, so every voxel (an MFMA row) is scaled by its own largest |go| -- exact to fp32 relative to that voxel's output gradient
typedef _Float16 dcn_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dcn_u32x4 __attribute__((ext_vector_type(4)));
#define DPF_STAMP_WAVE wave
template <int NST, int NW, bool F16>
__global__ __launch_bounds__(64 * NW) void dcn_bwd_input_pk_kernel(const float* __restrict__ offset, const float* __restrict__ wt2 /*[T][64][CT], zero rows beyond K*/,
                                                                   const float* __restrict__ go, float* __restrict__ dx, DcnP p, GiP q, int CT,
                                                                   const float* __restrict__ wmaxv /*[chunks of GI_CH] max |W|*/,
                                                                   long long* gi_shadow /* deterministic mode: dcn_internal.h */, const int* __restrict__ wexp) {
  extern __shared__ __align__(16) long long smem_q[];
  constexpr int NT = 64 * NW;
  constexpr int npos = 16 * NST * NW;
  constexpr bool HALVES = NW >= 8;   // thread pair per voxel in the table phase (4 corners each: one z side)
  constexpr int FR = HALVES ? 2 : 1; // table threads per voxel = rows of the per-voxel far flags
  // threads that build tables: with 16 waves the first 8 (four threads per voxel -- two corners each -- measured slower, 8.86 vs 8.48 ms:
  // the per-voxel part of the corner arithmetic and the offset loads are repeated in every thread of a voxel)
  constexpr int TBL_T = FR * npos;
  const int regvox = q.RZmax * q.RY * q.RX;
  long long* s_regq = smem_q;                                    // [regvox + 1][PK_CS]: cell = 8 packed channel pairs; last cell = dummy
  int* s_lidx = (int*)(s_regq + (size_t)(regvox + 1) * PK_CS);   // [2][npos][8] u64 index of the corner's cell (dummy cell when outside)
  float* s_w = (float*)(s_lidx + 2 * npos * 8);                  // [2][npos][8] corner weight, 0 outside the region
  unsigned* s_mass = (unsigned*)(s_w + 2 * npos * 8);            // [regvox + 4] weight mass per cell, PK_MASS_Q fixed point
  int* s_far = (int*)(s_mass + ((regvox + 4) & ~3));             // [4] flag of tap t in slot t % 3: some corner left the region
  int* s_farm = s_far + 4;                                       // [2][FR][npos] per-voxel flag (one row per table thread of a voxel)
  float* s_gmax = (float*)(s_farm + 2 * FR * npos);              // [NW]
  unsigned* s_mmax = (unsigned*)(s_gmax + NW);                   // [NW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int pr = l15 >> 1;           // channel pair within the 16-channel chunk
  const int jb = (l15 & 1) * 4;      // this lane's corners: 0-3 (even lane) or 4-7 (odd lane)
  const bool odd = (l15 & 1) != 0;
  // the two waves that share a SIMD (wave, wave + 4) run the two halves of a tap step in opposite order: one builds the next tap's
  // tables (vector ALU) while the other feeds the LDS atomic unit and the matrix pipe
  const bool tables_first = __builtin_amdgcn_readfirstlane(NW >= 8 ? (wave >> 2) & 1 : wave & 1) != 0;

  int bb = dpf_xcd_tile(blockIdx.x, gridDim.x);
  const int tx = bb % q.tilesX; bb /= q.tilesX;
  const int ty = bb % q.tilesY; bb /= q.tilesY;
  const int tz = bb % q.tilesZ;
  const int b = bb / q.tilesZ;
  const int z0 = tz * q.TZ, y0 = ty * GI_TY, x0 = tx * GI_TX;
  const int rz0u = z0 * p.sd - p.pd - GI_R, ry0 = y0 * p.sh - p.ph - GI_R, rx0 = x0 * p.sw - p.pw - GI_R;
  const int rz0 = rz0u < 0 ? 0 : rz0u;
  int rz1 = rz0u + (q.TZ - 1) * p.sd + (p.kd - 1) * p.dd + 1 + 2 * GI_R;
  if (rz1 > p.D) rz1 = p.D;
  int RZ = rz1 - rz0;
  if (RZ > q.RZmax) RZ = q.RZmax;

  const long long chan = (long long)p.D * p.H * p.W;
  const float* off_b = offset + (long long)b * 3 * p.T * p.P;
  float* dxb = dx + (long long)b * p.C * chan;

  const int vox = FR > 1 ? (tid & (npos - 1)) : tid;       // this thread's voxel in the table phase
  const int half = FR > 1 ? (tid / npos) & 1 : 0;          // its z side
  const bool tbl_thread = tid < TBL_T;                     // wave-uniform
  const int pdx = vox & (GI_TX - 1), pdy = (vox >> GI_TXS) & (GI_TY - 1), pdz = vox >> 6;
  const int zo = z0 + pdz, yo = y0 + pdy, xo = x0 + pdx;
  const bool pvalid = tbl_thread && pdz < q.TZ && zo < p.Do && yo < p.Ho && xo < p.Wo;
  const long long ppos = pvalid ? ((long long)zo * p.Ho + yo) * p.Wo + xo : p.P;

  // A fragments: go[k][voxel] for this wave's NST sub-tiles, all k (K <= 64 -> 16 k-steps of 4), kept in registers
  float afrag[NST][16];
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int pl = (wave * NST + st) * 16 + l15;
    const int ax = pl & (GI_TX - 1), ay = (pl >> GI_TXS) & (GI_TY - 1), az = pl >> 6;
    const int gz = z0 + az, gy = y0 + ay, gx = x0 + ax;
    const bool ok = az < q.TZ && gz < p.Do && gy < p.Ho && gx < p.Wo;
    const long long gpos = ((long long)gz * p.Ho + gy) * p.Wo + gx;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = F16 ? 32 * (ks >> 3) + 8 * lg + (ks & 7) : 4 * ks + lg;     // (F16: 8 consecutive k per lane and k half)
      afrag[st][ks] = (ok && k < p.K) ? go[((long long)b * p.K + k) * p.P + gpos] : 0.f;
    }
  }

  float gbound = 0.f;   // max over this workgroup's voxels of sum_k |go[k][voxel]|
  dcn_u32x4 aq[F16 ? NST : 1][2][2];                             // F16: [sub-tile][k half][hi | lo] of go * 2^(141 - exponent of the voxel)
  unsigned egp[F16 ? NST : 1];                                   // F16: byte r = exponent of voxel 4 lg + r (accumulator row r of this lane)
  int Ew = DPF_H3_EMIN;
  {
    float m = 0.f;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      float sa = 0.f, ma = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { sa += fabsf(afrag[st][ks]); ma = fmaxf(ma, fabsf(afrag[st][ks])); }
      sa += __shfl_xor(sa, 16, 64);
      sa += __shfl_xor(sa, 32, 64);
      m = fmaxf(m, sa);
      if constexpr (F16) {                                        // this lane's voxel (l15): all its output channels sit in lanes l15 + 16 lg'
        ma = fmaxf(ma, __shfl_xor(ma, 16, 64));
        ma = fmaxf(ma, __shfl_xor(ma, 32, 64));
        int e = (int)(__builtin_bit_cast(unsigned, ma) >> 23);
        e = e < DPF_H3_EMIN ? DPF_H3_EMIN : (e > 254 ? 254 : e);
        const float scg = dpf_h3_scale(e);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned h, l;
            dpf_split_pair_h(afrag[st][8 * mf + 2 * i] * scg, afrag[st][8 * mf + 2 * i + 1] * scg, h, l);
            aq[st][mf][0][i] = h; aq[st][mf][1][i] = l;
          }
        unsigned pk = 0;                                          // the exponents of this lane's accumulator rows (voxels 4 lg + r): lanes 4 lg + r hold them
#pragma unroll
        for (int r = 0; r < 4; ++r) pk |= (unsigned)__shfl(e, 4 * lg + r, 64) << (8 * r);
        egp[st] = pk;
      }
    }
    m = dpf_wave_max(m);
    if (lane == 0) s_gmax[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) gbound = fmaxf(gbound, s_gmax[w]);
    if constexpr (F16) Ew = __builtin_amdgcn_readfirstlane(wexp[0]);
  }

  const int zb = zo * p.sd - p.pd, yb = yo * p.sh - p.ph, xbase = xo * p.sw - p.pw;
  const float* offp0 = off_b + (pvalid ? ppos : 0);
  const int dummy = regvox * PK_CS;
  float massb = PK_MASS0;      // bound on the weight mass of a cell that the current pass is scaled for
  bool need_mass = true;       // the running pass also measures the mass
  int attempt = 0;             // 1: this pass is a repeat (far corners already emitted)

  // corner tables of one tap for this thread's voxel (its 4 or 8 corners) into table buffer `buf`.  Separable: per axis and side
  // (low / high corner) the in-volume flag, the in-region flag, the cell-index term and the linear weight; a corner is then two
  // adds, two multiplies and two mask ANDs (cuh:43-68 for the index / weight, cuh:248 for `valid`)
  const int RYX = q.RY * q.RX * PK_CS, RXC = q.RX * PK_CS;
  auto build_table = [&](int t, int buf, const TapIt& itc, const Off3& ocur) {
    if (tbl_thread) {
      const Corner cn = corner_at(p, pvalid, zb, yb, xbase, itc, ocur);
      bool okz[2], oky[2], okx[2], inz[2], iny[2], inx[2];
      int cz[2], cy[2], cx[2];
      float wz[2], wy[2], wx[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int d = cn.d0 + e, h = cn.h0 + e, w = cn.w0 + e;
        const int lz = d - rz0, ly = h - ry0, lx = w - rx0;
        okz[e] = cn.valid && d >= 0 && d <= p.D - 1;
        oky[e] = h >= 0 && h <= p.H - 1;
        okx[e] = w >= 0 && w <= p.W - 1;
        inz[e] = lz >= 0 && lz < RZ;
        iny[e] = ly >= 0 && ly < q.RY;
        inx[e] = lx >= 0 && lx < q.RX;
        cz[e] = lz * RYX; cy[e] = ly * RXC; cx[e] = lx * PK_CS;
        wz[e] = e ? cn.ld : 1.f - cn.ld;
        wy[e] = e ? cn.lh : 1.f - cn.lh;
        wx[e] = e ? cn.lw : 1.f - cn.lw;
      }
      bool anyfar = false;
      int liv[HALVES ? 4 : 8];
      float wv[HALVES ? 4 : 8];
#pragma unroll
      for (int jj = 0; jj < (HALVES ? 4 : 8); ++jj) {
        const int jd = HALVES ? -1 : (jj >> 2) & 1, jh = (jj >> 1) & 1, jw = jj & 1;
        // HALVES: this thread's z side is `half` (a run-time value): select the z terms once, outside (below)
        const bool ok = (HALVES ? (half ? okz[1] : okz[0]) : okz[jd]) && oky[jh] && okx[jw];
        const bool in = ok && (HALVES ? (half ? inz[1] : inz[0]) : inz[jd]) && iny[jh] && inx[jw];
        const int zc = HALVES ? (half ? cz[1] : cz[0]) : cz[jd];
        const float zw = HALVES ? (half ? wz[1] : wz[0]) : wz[jd];
        const float wg = zw * wy[jh] * wx[jw];
        liv[jj] = in ? zc + cy[jh] + cx[jw] : dummy;
        wv[jj] = in ? wg : 0.f;
        anyfar |= ok && !in;
        if (need_mass && in) atomicAdd(&s_mass[liv[jj] / PK_CS], (unsigned)__float2int_rn(wg * PK_MASS_Q));
      }
      int* lp = &s_lidx[(buf * npos + vox) * 8 + 4 * half];
      float* wp = &s_w[(buf * npos + vox) * 8 + 4 * half];
      *reinterpret_cast<int4*>(lp) = make_int4(liv[0], liv[1], liv[2], liv[3]);
      *reinterpret_cast<float4*>(wp) = make_float4(wv[0], wv[1], wv[2], wv[3]);
      if (!HALVES) {
        *reinterpret_cast<int4*>(lp + 4) = make_int4(liv[4], liv[5], liv[6], liv[7]);
        *reinterpret_cast<float4*>(wp + 4) = make_float4(wv[4], wv[5], wv[6], wv[7]);
      }
      s_farm[(buf * FR + half) * npos + vox] = anyfar ? 1 : 0;
      if (anyfar) s_far[t % 3] = 1;
    }
  };

  for (int c0 = 0; c0 < q.CG;) {      // only the channels whose gradient the caller needs
    __syncthreads();                                            // previous pass flushed
    float qscale = 0.f, qinv = 0.f;
    {
      float wm = wmaxv[c0 / GI_CH];
      if (c0 + GI_CH < p.C) wm = fmaxf(wm, wmaxv[c0 / GI_CH + 1]);
      const float B = gbound * wm * massb;
      if (B > 0.f) { qscale = 1073741824.f / B; qinv = B * (1.f / 1073741824.f); }
    }
    for (int i = tid; i < (regvox + 1) * PK_CS; i += NT) s_regq[i] = 0;
    if (need_mass)
      for (int i = tid; i < regvox + 1; i += NT) s_mass[i] = 0u;
    if (tid < 3) s_far[tid] = 0;
    const int cc = c0 + l15;                                    // this lane's gcol column (B fragment); < CT always
    const int ce = c0 + 2 * pr;                                 // even channel of the pair this lane scatters
    const float* offp = offp0;
    Off3 onext = load_off_ptr(offp, p.P, pvalid);
    TapIt it = {0, 0, 0};
    float bnext[F16 ? 1 : 16];
    dcn_u32x4 bq[2][2];                                          // F16: [k half][hi | lo]
    const float* wtn = wt2 + (long long)lg * CT + cc;           // rows k >= K of the repacked tensor are zero
    const char* wqn = reinterpret_cast<const char*>(wt2) + (long long)(c0 / PK_CH) * 4096 + lane * 16;     // F16: 4 KB per (tap, chunk)
    const long long wq_step = (long long)(CT / PK_CH) * 4096;
    if constexpr (F16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bq[u >> 1][u & 1] = *reinterpret_cast<const dcn_u32x4*>(wqn + u * 1024);
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bnext[ks] = wtn[(4 * ks) * CT];
    }
    __syncthreads();                                            // region / flags cleared
    {                                                           // tables of tap 0
      const Off3 o0 = onext;
      offp += 3 * p.P;
      onext = load_off_ptr(offp, p.P, pvalid && 1 < p.T);
      build_table(0, 0, it, o0);
    }
    TapIt it_cur = it;                                          // tap t; `it` runs one tap ahead (the tables being built)
    tap_next(p, it);
    __syncthreads();
    for (int t = 0; t < p.T; ++t) {
      const int cur = t & 1;
      if (c0 == PK_CH) { DPF_STAMP(2 * t, 0) }
      if (tid == 0) s_far[(t + 2) % 3] = 0;                     // slot of tap t + 2: nobody reads or sets it during this step
      const Off3 ocur = onext;                                  // offsets of tap t + 1
      offp += 3 * p.P;
      if (tbl_thread) {
        const float* np = t + 2 < p.T ? offp : offp0;             // always inside the tensor: unconditional loads
        onext = Off3{np[0], np[p.P], np[2 * p.P]};
      }
      if (tables_first && t + 1 < p.T) build_table(t + 1, cur ^ 1, it, ocur);
      if (c0 == PK_CH) { DPF_STAMP(2 * t, 1) }
      // ---- scatter of tap t
      const int* lidx = s_lidx + cur * npos * 8;
      const float* wtab = s_w + cur * npos * 8;
      f32x4 acc[NST];
#pragma unroll
      for (int st = 0; st < NST; ++st) acc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
      // B fragments: W[k][c0 + l15][t], prefetched one tap ahead; the next tap's loads are issued once the chain has consumed these
      // (the registers are reused: the 16-wave variant has 128)
      if constexpr (F16) {
        constexpr int ca[3] = {1, 0, 0}, cb[3] = {0, 1, 0};      // lo*hi, hi*lo, hi*hi
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int st = 0; st < NST; ++st)
              acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dcn_f16x8, aq[st][mf][ca[i]]), __builtin_bit_cast(dcn_f16x8, bq[mf][cb[i]]),
                                                               acc[st], 0, 0, 0);
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[st][r] = __builtin_ldexpf(acc[st][r], (int)((egp[st] >> (8 * r)) & 0xffu) + Ew - 282);
        if (t + 1 < p.T) {
          wqn += wq_step;
#pragma unroll
          for (int u = 0; u < 4; ++u) bq[u >> 1][u & 1] = *reinterpret_cast<const dcn_u32x4*>(wqn + u * 1024);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
          for (int st = 0; st < NST; ++st) acc[st] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[st][ks], bnext[ks], acc[st], 0, 0, 0);
        if (t + 1 < p.T) {
          wtn += 64 * CT;
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) bnext[ks] = wtn[(4 * ks) * CT];
        }
      }
      const bool far_tap = attempt == 0 && s_far[t % 3] != 0;   // block-uniform: some corner of this tap left the region
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        float ge[4], gd[4];    // gcol of the pair's even / odd channel for the 4 voxels (D rows) of this lane group
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mine = acc[st][r];
          const float other = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mine), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]: lane ^ 1
          ge[r] = odd ? other : mine;
          gd[r] = odd ? mine : other;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pl = (wave * NST + st) * 16 + 4 * lg + r;   // D row = voxel
          const int4 la = *reinterpret_cast<const int4*>(&lidx[pl * 8 + jb]);
          const float4 wa = *reinterpret_cast<const float4*>(&wtab[pl * 8 + jb]);
          const int li[4] = {la.x, la.y, la.z, la.w};
          const float wv[4] = {wa.x, wa.y, wa.z, wa.w};
          const float es = ge[r] * qscale, os = gd[r] * qscale;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int qe = cvt_rpi(es * wv[j]), qo = cvt_rpi(os * wv[j]);
            const long long pk = ((long long)qe << 32) + (long long)qo;
            atomicAdd(reinterpret_cast<unsigned long long*>(&s_regq[li[j] + pr]), (unsigned long long)pk);
          }
        }
        if (far_tap) {
          // rare: corners outside the staged box go straight to global memory; their tables are not kept -- the corner is recomputed
          // from the voxel's offsets
#pragma unroll 1
          for (int r = 0; r < 4; ++r) {
            const int pl = (wave * NST + st) * 16 + 4 * lg + r;
            int anyf = 0;
#pragma unroll
            for (int fr = 0; fr < FR; ++fr) anyf |= s_farm[(cur * FR + fr) * npos + pl];
            if (anyf == 0) continue;
            const int fx = pl & (GI_TX - 1), fy = (pl >> GI_TXS) & (GI_TY - 1), fz = pl >> 6;
            const int fzo = z0 + fz, fyo = y0 + fy, fxo = x0 + fx;
            const long long fpos = ((long long)fzo * p.Ho + fyo) * p.Wo + fxo;
            const Off3 fo = load_off_ptr(off_b + (long long)(3 * t) * p.P + fpos, p.P, true);
            const Corner cn = corner_at(p, true, fzo * p.sd - p.pd, fyo * p.sh - p.ph, fxo * p.sw - p.pw, it_cur, fo);
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
              float wg;
              const int v = corner_index32(p, cn, jb + j, wg);
              if (v < 0) continue;
              const int jd = ((jb + j) >> 2) & 1, jh = ((jb + j) >> 1) & 1, jw = (jb + j) & 1;
              const int lz = cn.d0 + jd - rz0, ly = cn.h0 + jh - ry0, lx = cn.w0 + jw - rx0;
              if (lz >= 0 && lz < RZ && ly >= 0 && ly < q.RY && lx >= 0 && lx < q.RX) continue;   // went into the region
              if (ce < q.CG) dcn_acc_add(dx, gi_shadow, &dxb[(long long)ce * chan + v], wg * ge[r]);
              if (ce + 1 < q.CG) dcn_acc_add(dx, gi_shadow, &dxb[(long long)(ce + 1) * chan + v], wg * gd[r]);
            }
          }
        }
      }
      if (c0 == PK_CH) { DPF_STAMP(2 * t + 1, 0) }
      if (!tables_first && t + 1 < p.T) build_table(t + 1, cur ^ 1, it, ocur);
      it_cur = it;
      tap_next(p, it);
      if (c0 == PK_CH) { DPF_STAMP(2 * t + 1, 1) }
      __syncthreads();                                          // tables of tap t + 1 complete, those of tap t consumed
    }
    if (need_mass) {      // was the pass scaled for enough mass?
      unsigned m = 0u;
      for (int i = tid; i < regvox; i += NT) m = max(m, s_mass[i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
      if (lane == 0) s_mmax[wave] = m;
      __syncthreads();
      m = 0u;
#pragma unroll
      for (int w = 0; w < NW; ++w) m = max(m, s_mmax[w]);
      const float measured = (float)m * (1.001f / PK_MASS_Q) + 0.05f;
      need_mass = false;
      if (measured > massb) {      // repeat this pass with the measured bound (block-uniform branch)
        massb = measured;
        attempt = 1;
        continue;
      }
      massb = fmaxf(measured, 1.f);
    }
    // flush: a unit = 16 consecutive cells = 1 KB of LDS read as one conflict-free ds_read_b128 per lane (lane = (cell, quarter cell):
    // 4 channels); the global atomics of a quarter run along x (64-byte segments per channel plane)
    const int ncell = RZ * q.RY * q.RX;
    for (int u = wave; u * 16 < ncell; u += NW) {
      const int cell = u * 16 + (lane >> 2), qt = lane & 3;
      if (cell >= ncell) continue;
      const int lx = cell % q.RX, zy = cell / q.RX;
      const int ly = zy % q.RY, lz = zy / q.RY;
      const int gz = rz0 + lz, gy = ry0 + ly, gx = rx0 + lx;
      if (gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) continue;
      const long long t0 = s_regq[cell * PK_CS + 2 * qt], t1 = s_regq[cell * PK_CS + 2 * qt + 1];
      const int lo0 = (int)t0, lo1 = (int)t1;
      const int hi0 = (int)((t0 - (long long)lo0) >> 32), hi1 = (int)((t1 - (long long)lo1) >> 32);
      const float v[4] = {(float)hi0 * qinv, (float)lo0 * qinv, (float)hi1 * qinv, (float)lo1 * qinv};
      float* dst = dxb + (long long)(c0 + 4 * qt) * chan + ((long long)gz * p.H + gy) * p.W + gx;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c0 + 4 * qt + k < q.CG && v[k] != 0.f) dcn_acc_add(dx, gi_shadow, dst + (long long)k * chan, v[k]);
    }
    attempt = 0;
    c0 += PK_CH;
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

// ==========================================================================================================
// Region-staged kernels.  A workgroup owns TZ x 2 x 32 output voxels (TZ = min(Do, 4)); the haloed input region
// [RZ][RY][RX] (halo g.R on top of the kernel extent) of a 16-channel chunk is staged in LDS once and all 27 taps sample it
// with ds_reads (8 per sample) instead of 8 scattered global loads; samples whose 2x2x2 corner block leaves the staged box
// take a global-memory slow path.
constexpr int RG_TY = 2, RG_TX = 32;
// Channels per chunk CH and floats per voxel VS of the channel-last LDS image [RV][VS]:
//   CH = 16, VS = 16: a lane reads the channels of a corner as 4 ds_read_b128; the four 16-byte quads of voxel v are stored at
//            quad position q ^ ((v >> 2) & 3) (XOR swizzle), so 16 consecutive voxels reading the same quad hit 16 different
//            16-byte slots of the 256-byte bank row (conflict free) without the 25 % padding a 20-float stride would cost.
//   CH = 12, VS = 12: for channel counts that 16 divides badly (35 -> 36 instead of 48 padded channels); 3*vox mod 16 is a
//            permutation too.  The smaller image leaves room for a halo of 4 instead of 3 (fewer samples on the slow path).
template <int CH>
struct RegCfg {
  static_assert(CH == 16 || CH == 12, "chunk width");
  static constexpr int VS = CH;
  // float offset of 16-byte quad q of region voxel v
  static __device__ __forceinline__ int quad(int v, int q) { return CH == 16 ? v * 16 + 4 * (q ^ ((v >> 2) & 3)) : v * 12 + 4 * q; }
};

struct RegGeo {
  int R;                   // halo on top of the kernel extent
  int RYH;                 // halo along y (>= R: rows are cheap, the global-memory slow path of a sample that leaves the region is not)
  int RXL;                 // left halo along x (>= R; chosen so that the region's first column is 16-byte aligned in global memory)
  int TZ, RZmax, RY, RX, RV;
  int tilesZ, tilesY, tilesX;
  int TX;                  // tile width in output voxels (32; 16 for the half-width forward tile; 8 for the compact 4 x 8 x 8 tile)
  int TY, TXS;             // tile rows (2; 8 for the compact tile) and log2(TX)
};

struct RegCtx {
  int b, z0, y0, x0;       // tile origin (output coordinates)
  int rz0, ry0, rx0, RZ;   // region origin (input coordinates) and clipped depth
};

__device__ __forceinline__ RegCtx region_ctx(const DcnP& p, const RegGeo& g, int blk) {
  RegCtx c;
  blk = dpf_xcd_tile(blk, gridDim.x);
  const int tx = blk % g.tilesX; blk /= g.tilesX;
  const int ty = blk % g.tilesY; blk /= g.tilesY;
  const int tz = blk % g.tilesZ;
  c.b = blk / g.tilesZ;
  c.z0 = tz * g.TZ; c.y0 = ty * g.TY; c.x0 = tx * g.TX;
  const int rz0u = c.z0 * p.sd - p.pd - g.R;
  c.rz0 = rz0u < 0 ? 0 : rz0u;
  int rz1 = rz0u + (g.TZ - 1) * p.sd + (p.kd - 1) * p.dd + 1 + 2 * g.R;
  if (rz1 > p.D) rz1 = p.D;
  c.RZ = rz1 - c.rz0;
  if (c.RZ > g.RZmax) c.RZ = g.RZmax;
  c.ry0 = c.y0 * p.sh - p.ph - g.RYH;
  c.rx0 = c.x0 * p.sw - p.pw - g.RXL;
  return c;
}

// stage x[b, c0 .. c0+nch) over the region into s_reg[ch][RV] (zeros outside the volume / beyond C)
template <int CH>
__device__ __forceinline__ void stage_region(const DcnP& p, const RegGeo& g, const RegCtx& c, const float* __restrict__ xb, int c0, float* s_reg,
                                             int wave_u, int lane, int nwaves = 4) {
  const long long chan = (long long)p.D * p.H * p.W;
  const int rows_per_ch = c.RZ * g.RY;
  const int nrows = CH * rows_per_ch;
  constexpr int SU = 8;
  for (int r0 = wave_u * SU; r0 < nrows; r0 += nwaves * SU) {
    float v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int row = r0 + u;
      const int ch = row / rows_per_ch;
      const int rem = row - ch * rows_per_ch;
      const int lz = rem / g.RY, ly = rem - lz * g.RY;
      const int gz = c.rz0 + lz, gy = c.ry0 + ly, gx = c.rx0 + lane;
      const bool ok = row < nrows && (c0 + ch) < p.C && gy >= 0 && gy < p.H && lane < g.RX && gx >= 0 && gx < p.W;
      v[u] = ok ? xb[(long long)(c0 + ch) * chan + ((long long)gz * p.H + gy) * p.W + gx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int row = r0 + u;
      if (row < nrows && lane < g.RX) {
        const int ch = row / rows_per_ch;
        const int rem = row - ch * rows_per_ch;
        s_reg[RegCfg<CH>::quad(rem * g.RX + lane, ch >> 2) + (ch & 3)] = v[u];
      }
    }
  }
}

// Vectorised staging (needs W % 4 == 0, rx0 % 4 == 0, RX % 4 == 0, 16-byte aligned x): a unit = 4 channels x 4 consecutive x of
// one region row, fetched as 4 float4 loads (one per channel), transposed in registers and written as 4 ds_write_b128 (one
// per voxel of the channel-last image) -- 6x fewer load and 16x fewer LDS-write instructions than stage_region, all loads of
// a thread in flight together.
template <int CH>
__device__ __forceinline__ void stage_region4(const DcnP& p, const RegGeo& g, const RegCtx& c, const float* __restrict__ xb, int c0, float* s_reg,
                                              int tid, int nthreads) {
  constexpr int NCG = CH / 4;
  const long long chan = (long long)p.D * p.H * p.W;
  const int SR = g.RX >> 2;
  const int units = c.RZ * g.RY * NCG * SR;
  for (int u = tid; u < units; u += nthreads) {
    const int seg = u % SR;
    const int it = u / SR;
    const int cg = it % NCG;
    const int row = it / NCG;
    const int lz = row / g.RY, ly = row - lz * g.RY;
    const int gz = c.rz0 + lz, gy = c.ry0 + ly, gx = c.rx0 + 4 * seg;
    const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const float* src = xb + (long long)(c0 + 4 * cg) * chan + ((long long)gz * p.H + gy) * p.W + gx;
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (ok && c0 + 4 * cg + q < p.C) ? *reinterpret_cast<const float4*>(src + q * chan) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int v0 = row * g.RX + 4 * seg;            // a multiple of 4: the four voxels share one swizzle key
    *reinterpret_cast<float4*>(s_reg + RegCfg<CH>::quad(v0, cg)) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
    *reinterpret_cast<float4*>(s_reg + RegCfg<CH>::quad(v0 + 1, cg)) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
    *reinterpret_cast<float4*>(s_reg + RegCfg<CH>::quad(v0 + 2, cg)) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
    *reinterpret_cast<float4*>(s_reg + RegCfg<CH>::quad(v0 + 3, cg)) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
  }
}

// picks the vectorised staging when the geometry allows it
template <int CH>
__device__ __forceinline__ void stage_region_any(const DcnP& p, const RegGeo& g, const RegCtx& c, const float* __restrict__ xb, int c0, float* s_reg,
                                                 int tid, int nthreads, bool vec) {
  if (vec) {
    stage_region4<CH>(p, g, c, xb, c0, s_reg, tid, nthreads);
  } else {
    stage_region<CH>(p, g, c, xb, c0, s_reg, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, nthreads >> 6);
  }
}

struct Samp {
  int base;            // LDS index of the (low,low,low) corner (clamped), fast path only
  int dzs;             // LDS stride to the high-z corner (0 when clamped)
  float wz[2], wy[2], wx[2];   // trilinear factors with the in-volume masks folded in
  float mz[2], my[2], mx[2];   // in-volume masks (for the coordinate derivatives)
  bool valid, fast;
};

// Branch-free variant for the role-split samplers: a sample outside the volume (or a voxel outside the tile) is a FAST sample with all
// masks -- hence all weights and derivative factors -- zero that reads region cell 0: one exec-mask branch less in the step loop.
__device__ __forceinline__ Samp make_samp_nb(const DcnP& p, const RegGeo& g, const RegCtx& c, const Corner& cn) {
  Samp s;
  const bool v = cn.valid != 0;
  const bool zl = v && cn.d0 >= 0 && cn.d0 <= p.D - 1, zh = v && cn.d0 + 1 >= 0 && cn.d0 + 1 <= p.D - 1;
  const bool yl = v && cn.h0 >= 0 && cn.h0 <= p.H - 1, yh = v && cn.h0 + 1 >= 0 && cn.h0 + 1 <= p.H - 1;
  const bool xl = v && cn.w0 >= 0 && cn.w0 <= p.W - 1, xh = v && cn.w0 + 1 >= 0 && cn.w0 + 1 <= p.W - 1;
  s.mz[0] = zl ? 1.f : 0.f; s.mz[1] = zh ? 1.f : 0.f;
  s.my[0] = yl ? 1.f : 0.f; s.my[1] = yh ? 1.f : 0.f;
  s.mx[0] = xl ? 1.f : 0.f; s.mx[1] = xh ? 1.f : 0.f;
  s.wz[0] = (1.f - cn.ld) * s.mz[0]; s.wz[1] = cn.ld * s.mz[1];
  s.wy[0] = (1.f - cn.lh) * s.my[0]; s.wy[1] = cn.lh * s.my[1];
  s.wx[0] = (1.f - cn.lw) * s.mx[0]; s.wx[1] = cn.lw * s.mx[1];
  const int lz = cn.d0 - c.rz0, ly = cn.h0 - c.ry0, lx = cn.w0 - c.rx0;
  const bool yx_in = ly >= 0 && ly + 1 < g.RY && lx >= 0 && lx + 1 < g.RX;
  const bool zlo_in = lz >= 0 && lz < c.RZ, zhi_in = lz + 1 >= 0 && lz + 1 < c.RZ;
  const bool inreg = v && yx_in && (zlo_in || !zl) && (zhi_in || !zh);
  const int iz0 = zlo_in ? lz : (zhi_in ? lz + 1 : 0);
  const int iz1 = zhi_in ? lz + 1 : iz0;
  s.base = inreg ? (iz0 * g.RY + ly) * g.RX + lx : 0;
  s.dzs = inreg ? (iz1 - iz0) * g.RY * g.RX : 0;
  s.valid = true;
  s.fast = inreg || !v;
  return s;
}

// 16 channels of corner (jd, jh, jw): 4 x ds_read_b128 from the channel-last image (fast path)
template <int CH>
__device__ __forceinline__ void corner_vec(const RegGeo& g, const Samp& s, const float* s_reg, int jd, int jh, int jw, float v[CH]) {
  const int vx = s.base + jd * s.dzs + jh * g.RX + jw;
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const float4 f = *reinterpret_cast<const float4*>(s_reg + RegCfg<CH>::quad(vx, q));
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
}

constexpr int ST = 256 + 4;   // padded row of the [16][256] sample / gcol tile

// ---- 8-wave forward (two waves per SIMD on the same LDS image): a thread pair per voxel, each thread sampling half of the
// chunk's channels; each wave contracts 32 voxels.
template <int CH, int H>
__device__ __forceinline__ void corner_half(const RegGeo& g, const Samp& s, const float* s_reg, int jd, int jh, int jw, float v[CH / 2]) {
  const int vx = s.base + jd * s.dzs + jh * g.RX + jw;
  const float* r = s_reg + vx * RegCfg<CH>::VS + H * (CH / 2);      // CH = 12: linear image
  if (CH == 16) {
    const float4 a = *reinterpret_cast<const float4*>(s_reg + RegCfg<CH>::quad(vx, 2 * H)),
                 b = *reinterpret_cast<const float4*>(s_reg + RegCfg<CH>::quad(vx, 2 * H + 1));
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if (H == 0) {
    const float4 a = *reinterpret_cast<const float4*>(r);
    const float2 b = *reinterpret_cast<const float2*>(r + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y;
  } else {
    const float2 a = *reinterpret_cast<const float2*>(r);
    const float4 b = *reinterpret_cast<const float4*>(r + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = b.z; v[5] = b.w;
  }
}

// ---- role-split forward: waves 0-3 SAMPLE (one output voxel per thread, all CH channels of the staged chunk: VALU + LDS reads),
// waves 4-7 CONTRACT (64 voxels per wave on the fp32 matrix cores).  The sampled tile S[CH][256] is double buffered, so the
// samplers build tap t+1 while the MFMA waves consume tap t: one barrier per tap instead of two, and the vector / LDS pipes and the
// matrix pipe of every SIMD (which hosts one wave of each role) run concurrently.  The 4- and 8-wave kernels above alternate the
// two phases with every wave in lockstep at one workgroup per CU (LDS), which leaves each pipe idle for the other's phase.
// half-chunk sampler of the 16-wave forward kernel: writes its NC samples straight into the tile column `dst` (row stride STR).  Fast
// path: two straight-line groups of 4 corners (8 ds_read_b128 in flight); slow path (a corner outside the staged box): channel by
// channel from global memory, rolled -- it costs the fast path no registers.
template <int CH, int H, int ROWS>
__device__ __forceinline__ void fwd_sample_half_store(const DcnP& p, const RegGeo& g, const Samp& sp, const Corner& cn, const float* s_reg,
                                                      const float* __restrict__ xb, int c0, long long chan, float* dst) {
  constexpr int NC = CH / 2;
  float* dh = dst + H * NC * ROWS;
  if (sp.fast) {      // (samples outside the volume arrive as fast samples with zero weights: make_samp_nb)
    float val[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) val[ch] = 0.f;
#pragma unroll
    for (int jd = 0; jd < 2; ++jd) {
      float v[4][NC];
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) corner_half<CH, H>(g, sp, s_reg, jd, jy >> 1, jy & 1, v[jy]);
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) {
        const float wj = sp.wz[jd] * sp.wy[jy >> 1] * sp.wx[jy & 1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) val[ch] = fmaf(wj, v[jy][ch], val[ch]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) dh[ch * ROWS] = val[ch];
  } else {
    int vx[8];
    float wj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
      const int d = cn.d0 + jd, h = cn.h0 + jh, w = cn.w0 + jw;
      const bool in = d >= 0 && d <= p.D - 1 && h >= 0 && h <= p.H - 1 && w >= 0 && w <= p.W - 1;
      vx[j] = in ? (d * p.H + h) * p.W + w : -1;
      wj[j] = sp.wz[jd] * sp.wy[jh] * sp.wx[jw];
    }
#pragma unroll 1
    for (int ch = 0; ch < NC; ++ch) {
      const int cg = c0 + H * NC + ch;
      const float* xc = xb + (long long)(cg < p.C ? cg : p.C - 1) * chan;
      float sv = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = (vx[j] >= 0 && cg < p.C) ? xc[vx[j]] : 0.f;
        sv = fmaf(wj[j], xv, sv);
      }
      dh[ch * ROWS] = sv;
    }
  }
}

// full-chunk sampler of the 8-wave (4-wave at TXV = 16) forward kernel: all CH channels of one voxel, written straight into the tile
// column `dst`.  Same structure as the half sampler: straight-line fast path in two groups of 4 corners, rolled slow path.
template <int CH, int ROWS>
__device__ __forceinline__ void fwd_sample_store(const DcnP& p, const RegGeo& g, const Samp& sp, const Corner& cn, const float* s_reg,
                                                 const float* __restrict__ xb, int c0, long long chan, float* dst) {
  if (sp.fast) {      // (samples outside the volume arrive as fast samples with zero weights: make_samp_nb)
    float val[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) val[ch] = 0.f;
#pragma unroll
    for (int jd = 0; jd < 2; ++jd) {
      float v[4][CH];
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) corner_vec<CH>(g, sp, s_reg, jd, jy >> 1, jy & 1, v[jy]);
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) {
        const float wj = sp.wz[jd] * sp.wy[jy >> 1] * sp.wx[jy & 1];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) val[ch] = fmaf(wj, v[jy][ch], val[ch]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) dst[ch * ROWS] = val[ch];
  } else {
    int vx[8];
    float wj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
      const int d = cn.d0 + jd, h = cn.h0 + jh, w = cn.w0 + jw;
      const bool in = d >= 0 && d <= p.D - 1 && h >= 0 && h <= p.H - 1 && w >= 0 && w <= p.W - 1;
      vx[j] = in ? (d * p.H + h) * p.W + w : -1;
      wj[j] = sp.wz[jd] * sp.wy[jh] * sp.wx[jw];
    }
#pragma unroll 1
    for (int ch = 0; ch < CH; ++ch) {
      const int cg = c0 + ch;
      const float* xc = xb + (long long)(cg < p.C ? cg : p.C - 1) * chan;
      float sv = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = (vx[j] >= 0 && cg < p.C) ? xc[vx[j]] : 0.f;
        sv = fmaf(wj[j], xv, sv);
      }
      dst[ch * ROWS] = sv;
    }
  }
}

constexpr int STR = 256;   // row of the [CH][256] sample tile (lane-consecutive writes and reads: no padding needed)
// NW = 8: 4 sampler waves (a voxel per thread, all CH channels) + 4 MFMA waves (64 voxels each).
// NW = 16: 8 sampler waves (a thread PAIR per voxel, half the channels each) + 8 MFMA waves (32 voxels each): four waves per SIMD --
//          the kernel is latency bound at one LDS-limited workgroup per CU (measured: MFMA 25 %, VALU 17 %, LDS 20 % busy with 8 waves).
#undef DPF_STAMP_WAVE
#define DPF_STAMP_WAVE wave_u
// TXV: tile width (32: 4 x 2 x 32 = 256 voxels per workgroup; 16: 128 voxels, half the LDS image at the same halo ratio -- two workgroups
// per CU, out of phase with each other).  Samplers cover the tile once (a voxel per thread, all CH channels) or twice (thread pair
// per voxel, half the channels each) depending on NW.
template <int MT, int CH, int NW, int TXV>
__global__ __launch_bounds__(64 * NW) void dcn_fwd_rs_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                             const float* __restrict__ wt /*[T][Cpad][KT], zero rows beyond C*/,
                                                             const float* __restrict__ bias, float* __restrict__ out, DcnP p, RegGeo g, int vec) {
  extern __shared__ __align__(16) float smem[];
  constexpr int KT = 32 * MT;
  constexpr int NS = NW / 2;                          // sampler waves = MFMA waves
  constexpr int NVOX = 8 * TXV;                       // output voxels per workgroup
  constexpr int NTW = NVOX / (32 * NS);               // 32-voxel column tiles per MFMA wave (2 or 1)
  constexpr bool HALF = NS * 64 == 2 * NVOX;          // thread pair per voxel
  constexpr int STRV = NVOX;                          // row of the [CH][NVOX] sample tile
  static_assert(NTW == 1 || NTW == 2, "tile / wave split");
  float* s_reg = smem;                                // [RV][VS]
  float* s_S = s_reg + RegCfg<CH>::VS * g.RV;         // [2][CH][STRV]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RegCtx c = region_ctx(p, g, blockIdx.x);
  const long long chan = (long long)p.D * p.H * p.W;
  const float* xb = x + (long long)c.b * p.C * chan;
  // The role branch is the OUTERMOST construct: each role owns its chunk / tap loops, so no value of one role is live through the
  // other's code (with the branch inside the chunk loop the 16-wave variant spilled 45 registers in its MFMA loop).
  if (wave_u < NS) {
    // ------------------------------------------------------------------------------------------------ samplers
    const int vox = tid % NVOX;
    const int half = tid / NVOX;                       // HALF: the first NVOX threads take channels [0, CH/2), the others the rest
    const float* off_b = offset + (long long)c.b * 3 * p.T * p.P;
    const int pdx = vox & (g.TX - 1), pdy = (vox >> g.TXS) & (g.TY - 1), pdz = vox / (2 * TXV);   // g.TX * g.TY = 2 * TXV voxels per plane
    const int zo = c.z0 + pdz, yo = c.y0 + pdy, xo = c.x0 + pdx;
    const bool pvalid = pdz < g.TZ && zo < p.Do && yo < p.Ho && xo < p.Wo;
    const long long ppos = pvalid ? ((long long)zo * p.Ho + yo) * p.Wo + xo : p.P;
    const int zb = zo * p.sd - p.pd, yb = yo * p.sh - p.ph, xbase = xo * p.sw - p.pw;
    const float* offp0 = off_b + (pvalid ? ppos : 0);
#pragma unroll 1
    for (int c0 = 0; c0 < p.C; c0 += CH) {
      __syncthreads();                                 // samplers are done with the previous chunk's region
      stage_region_any<CH>(p, g, c, xb, c0, s_reg, tid, 64 * NW, vec != 0);
      const float* offp = offp0;
      Off3 onext = load_off_ptr(offp, p.P, pvalid);
      TapIt it = {0, 0, 0};
      __syncthreads();
#pragma unroll 1
      for (int t = 0; t < p.T; ++t) {
        if (c0 == CH) { DPF_STAMP(t, 0) }
        const Off3 ocur = onext;
        offp += 3 * p.P;
        {   // always-valid pointer, unconditional loads (a conditional load is a branch in the tap loop)
          const float* np = t + 1 < p.T ? offp : offp0;
          onext = Off3{np[0], np[p.P], np[2 * p.P]};
        }
        const Corner cn = corner_at(p, pvalid, zb, yb, xbase, it, ocur);
        tap_next(p, it);
        const Samp sp = make_samp_nb(p, g, c, cn);
        float* dst = s_S + (t & 1) * (CH * STRV) + vox;
        if (!HALF) {
          fwd_sample_store<CH, STRV>(p, g, sp, cn, s_reg, xb, c0, chan, dst);
        } else {
          if (half == 0) fwd_sample_half_store<CH, 0, STRV>(p, g, sp, cn, s_reg, xb, c0, chan, dst);
          else fwd_sample_half_store<CH, 1, STRV>(p, g, sp, cn, s_reg, xb, c0, chan, dst);
        }
        if (c0 == CH) { DPF_STAMP(t, 1) }
        __syncthreads();                               // barrier t: S[t&1] is complete; the MFMA waves have finished reading S[(t-1)&1]
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------ MFMA waves
    const int mw = wave_u - NS;
    const int Cpad = (p.C + CH - 1) / CH * CH;
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < p.C; c0 += CH) {
      __syncthreads();
      stage_region_any<CH>(p, g, c, xb, c0, s_reg, tid, 64 * NW, vec != 0);
      __syncthreads();
      // weight fragments of tap t are fetched (L2) one tap ahead
      float aN[CH / 2][MT];
      {
        const float* wtt = wt + ((long long)c0 + hh) * KT + l31;
#pragma unroll
        for (int sx = 0; sx < CH / 2; ++sx)
#pragma unroll
          for (int m = 0; m < MT; ++m) aN[sx][m] = wtt[(2 * sx) * KT + m * 32];
      }
#pragma unroll 1
      for (int t = 0; t < p.T; ++t) {
        float a[CH / 2][MT];
#pragma unroll
        for (int sx = 0; sx < CH / 2; ++sx)
#pragma unroll
          for (int m = 0; m < MT; ++m) a[sx][m] = aN[sx][m];
        if (t + 1 < p.T) {
          const float* wtt = wt + ((long long)(t + 1) * Cpad + c0 + hh) * KT + l31;
#pragma unroll
          for (int sx = 0; sx < CH / 2; ++sx)
#pragma unroll
            for (int m = 0; m < MT; ++m) aN[sx][m] = wtt[(2 * sx) * KT + m * 32];
        }
        __syncthreads();                               // barrier t
        if (c0 == CH) { DPF_STAMP(t, 0) }
        const float* src = s_S + (t & 1) * (CH * STRV) + mw * (32 * NTW) + l31;
#pragma unroll
        for (int sx = 0; sx < CH / 2; ++sx) {
          float bv[NTW];
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) bv[nt] = src[(2 * sx + hh) * STRV + nt * 32];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sx][m], bv[nt], acc[m][nt], 0, 0, 0);
        }
        if (c0 == CH) { DPF_STAMP(t, 1) }
      }
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int pl = mw * (32 * NTW) + nt * 32 + l31;
      const int ax = pl & (g.TX - 1), ay = (pl >> g.TXS) & (g.TY - 1), az = pl / (2 * TXV);
      const int gz = c.z0 + az, gy = c.y0 + ay, gx = c.x0 + ax;
      if (az < g.TZ && gz < p.Do && gy < p.Ho && gx < p.Wo) {
        const long long pos = ((long long)gz * p.Ho + gy) * p.Wo + gx;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
            if (k < p.K) out[((long long)c.b * p.K + k) * p.P + pos] = acc[m][nt][j] + (bias ? bias[k] : 0.f);
          }
      }
    }
  }
}

constexpr int WG_NREP = DCN_WG_NREP;   // replicas of the grad_weight scratch tensor (dcn_internal.h: shared with the lean kernels)

// ---- role-split grad_offset + grad_weight: 16 waves = 8 SAMPLER waves (a thread PAIR per voxel, each thread half of the chunk's
// channels: corner reads, coordinate derivatives, samples), 4 GCOL waves (gcol = W^T go for 64 voxels each) and 4 WGRAD waves (dW partial
// for 16 output channels each).  Every SIMD hosts two sampler waves and one wave of each MFMA role, so its matrix pipe (two MFMA
// streams), its vector ALU and the LDS pipe run concurrently; the fused 4-wave kernel above alternates the phases in lockstep at one
// wave per SIMD and holds both go layouts plus the sampling state in one 410-register wave (counters of a 4-sampler version: VALU 25 %,
// MFMA 28 %, LDS 19 % busy, 61 % of the wave cycles waiting -- one sampler wave per SIMD is a single dependent chain).  Here each MFMA
// role holds ONE layout of go (64 registers) and the samplers none, inside the 128-register budget of four waves per SIMD.  Steps
// are the flattened (channel chunk, tap) pairs; three [CH][256] tiles rotate:
//   step i:  GCOL writes gcol(i+1) into X[(i+1)%3] | SAMPLERS read gcol(i) from X[i%3], overwrite their own half column with the samples
//            S(i) | WGRAD contracts S(i-1) from X[(i-1)%3]                                  -- one barrier per step.
// The two halves of a voxel's coordinate gradient meet one step later: the upper-half thread leaves its partial in s_part[i&1], the
// lower-half thread adds it during step i+1 and does the read-modify-write of grad_offset (whose read was issued a step earlier).
// gcol(i+1) and the dW product do not touch the staged region, so the pipeline runs through the chunk boundaries; only the
// re-staging itself (all 1024 threads) is bracketed by barriers.
constexpr int XS = 260;    // padded row of a rotating tile (wgrad B reads: rows l15, 4 consecutive voxels per lane group)

// one sampler step of half H: partial coordinate gradient (gd, gh, gw) over this half's channels; the samples replace the gcol column.
// Fast path: the 8 corners in two straight-line groups of 4 (8 ds_read_b128 in flight).  Slow path (some corner outside the staged
// box, rare): channel by channel from global memory, rolled, so that it costs the fast path no registers.
template <int CH, int H>
__device__ __forceinline__ void rs_sample_half(const DcnP& p, const RegGeo& g, const Samp& sp, const Corner& cn, const float* s_reg,
                                               const float* __restrict__ xb, int c0, long long chan, float* col, float& gd, float& gh, float& gw) {
  constexpr int NC = CH / 2;
  gd = gh = gw = 0.f;
  float* colh = col + H * NC * XS;
  // (samples outside the volume arrive as fast samples with zero weights and masks: make_samp_nb)
  // dot_j = sum_ch gcol[ch] * x[corner j][ch]; the three coordinate derivatives weight it with the other two trilinear factors and the
  // signed in-volume mask of their own axis (cuh:131-187)
  float dots[8];
  if (sp.fast) {
    float gcv[NC], sval[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) { gcv[ch] = colh[ch * XS]; sval[ch] = 0.f; }   // gcol is zero for channels beyond C (zero weight columns)
#pragma unroll
    for (int jd = 0; jd < 2; ++jd) {
      float v[4][NC];
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) corner_half<CH, H>(g, sp, s_reg, jd, jy >> 1, jy & 1, v[jy]);
#pragma unroll
      for (int jy = 0; jy < 4; ++jy) {
        float dot = 0.f;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) dot = fmaf(gcv[ch], v[jy][ch], dot);
        dots[4 * jd + jy] = dot;
        const float wj = sp.wz[jd] * sp.wy[jy >> 1] * sp.wx[jy & 1];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) sval[ch] = fmaf(wj, v[jy][ch], sval[ch]);
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the second group's reads behind the first group's arithmetic (registers)
    }
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) colh[ch * XS] = sval[ch];
  } else {
    int vx[8];
    float wj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
      const int d = cn.d0 + jd, h = cn.h0 + jh, w = cn.w0 + jw;
      const bool in = d >= 0 && d <= p.D - 1 && h >= 0 && h <= p.H - 1 && w >= 0 && w <= p.W - 1;
      vx[j] = in ? (d * p.H + h) * p.W + w : -1;
      wj[j] = sp.wz[jd] * sp.wy[jh] * sp.wx[jw];
      dots[j] = 0.f;
    }
#pragma unroll 1
    for (int ch = 0; ch < NC; ++ch) {
      const int cg = c0 + H * NC + ch;
      const float* xc = xb + (long long)(cg < p.C ? cg : p.C - 1) * chan;
      const float gval = colh[ch * XS];
      float sv = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = (vx[j] >= 0 && cg < p.C) ? xc[vx[j]] : 0.f;
        dots[j] = fmaf(gval, xv, dots[j]);
        sv = fmaf(wj[j], xv, sv);
      }
      colh[ch * XS] = sv;
    }
  }
  // d/d(coord) of the trilinear sample, factored (34 instead of 72 operations): dots[4 jd + 2 jh + jw]
  {
    float A[2][2], E[2][2];      // A[jd][jh] = sum_jw wx dot;  E[jd][jw] = sum_jh wy dot
#pragma unroll
    for (int jd = 0; jd < 2; ++jd)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        A[jd][u] = fmaf(sp.wx[1], dots[4 * jd + 2 * u + 1], sp.wx[0] * dots[4 * jd + 2 * u]);
        E[jd][u] = fmaf(sp.wy[1], dots[4 * jd + 2 + u], sp.wy[0] * dots[4 * jd + u]);
      }
    const float B0 = fmaf(sp.wy[1], A[0][1], sp.wy[0] * A[0][0]), B1 = fmaf(sp.wy[1], A[1][1], sp.wy[0] * A[1][0]);
    gd = fmaf(sp.mz[1], B1, -sp.mz[0] * B0);
    const float C0 = fmaf(sp.wz[1], A[1][0], sp.wz[0] * A[0][0]), C1 = fmaf(sp.wz[1], A[1][1], sp.wz[0] * A[0][1]);
    gh = fmaf(sp.my[1], C1, -sp.my[0] * C0);
    const float F0 = fmaf(sp.wz[1], E[1][0], sp.wz[0] * E[0][0]), F1 = fmaf(sp.wz[1], E[1][1], sp.wz[0] * E[0][1]);
    gw = fmaf(sp.mx[1], F1, -sp.mx[0] * F0);
  }
}

#undef DPF_STAMP_WAVE
#define DPF_STAMP_WAVE wave_u
template <int CH>
__global__ __launch_bounds__(1024) void dcn_bwd_offset_rs_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                                 const float* __restrict__ wt2 /*[T][64][CT], zero rows beyond K*/,
                                                                 const float* __restrict__ go, float* __restrict__ doff, float* __restrict__ dwtmp,
                                                                 DcnP p, RegGeo g, int CT, int nchunk, int vec, int det) {
  extern __shared__ __align__(16) float smem[];
  float* s_reg = smem;                                 // [RV][VS]
  float* s_x = s_reg + RegCfg<CH>::VS * g.RV;          // [3][CH][XS]
  float* s_part = s_x + 3 * CH * XS;                   // [2][3][256] upper-half partials of (gd, gh, gw)
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave_u < 8 ? 0 : (wave_u < 12 ? 1 : 2);   // sampler, gcol, wgrad
  const int rw = wave_u & 3;
  const RegCtx c = region_ctx(p, g, blockIdx.x);
  const long long chan = (long long)p.D * p.H * p.W;
  const float* xb = x + (long long)c.b * p.C * chan;
  const int NS = nchunk * p.T;
  constexpr int XT = CH * XS;

  if (role == 0) {
    // ------------------------------------------------------------------------------------------------ samplers
    const int vox = tid & 255, half = wave_u >> 2;
    const float* off_b = offset + (long long)c.b * 3 * p.T * p.P;
    float* doff_b = doff + (long long)c.b * 3 * p.T * p.P;
    const int pdx = vox & (g.TX - 1), pdy = (vox >> g.TXS) & (g.TY - 1), pdz = vox >> 6;     // 64 voxels per plane: TX * TY = 64
    const int zo = c.z0 + pdz, yo = c.y0 + pdy, xo = c.x0 + pdx;
    const bool pvalid = pdz < g.TZ && zo < p.Do && yo < p.Ho && xo < p.Wo;
    const long long ppos = pvalid ? ((long long)zo * p.Ho + yo) * p.Wo + xo : p.P;
    const int zb = zo * p.sd - p.pd, yb = yo * p.sh - p.ph, xbase = xo * p.sw - p.pw;
    const float* offp0 = off_b + (pvalid ? ppos : 0);
    stage_region_any<CH>(p, g, c, xb, 0, s_reg, tid, 768, vec != 0);
    const float* offp = offp0;
    Off3 onext = load_off_ptr(offp, p.P, pvalid);
    __syncthreads();                                   // prologue: region of chunk 0 staged, gcol(0) written
    TapIt it = {0, 0, 0};
    int t = 0, c0 = 0;
    // lower half: the previous step's own partial and its grad_offset address.  The first channel chunk stores, the later ones add with
    // a no-return float atomic (same thread, same address, chunks apart in time): no read of the running value has to be waited for
    float pg[3] = {0.f, 0.f, 0.f};
    float* dqp = doff_b;
    bool first_chunk = true;
#pragma unroll 1
    for (int i = 0; i <= NS; ++i) {
      if (half == 0 && i >= 1 && pvalid) {      // finish step i - 1: both halves of the channel chunk
        const float* pp = s_part + ((i - 1) & 1) * 768 + vox;
        const float a0 = pg[0] + pp[0], a1 = pg[1] + pp[256], a2 = pg[2] + pp[512];
        if (first_chunk) { dqp[0] = a0; dqp[p.P] = a1; dqp[2 * p.P] = a2; }
        else { atomicAdd(dqp, a0); atomicAdd(dqp + p.P, a1); atomicAdd(dqp + 2 * p.P, a2); }
      }
      if (i == NS) break;
      DPF_STAMP(i, 0)
      first_chunk = c0 == 0;
      const Off3 ocur = onext;
      const bool last_tap = t + 1 == p.T;
      offp = last_tap ? offp0 : offp + 3 * p.P;
      // next step's offsets (tap 0 again after the last tap of a chunk).  The pointer is always inside the tensor (voxels outside the
      // volume read voxel 0 and are masked by pvalid later): unconditional loads -- a conditional one is a branch in the step loop
      onext = Off3{offp[0], offp[p.P], offp[2 * p.P]};
      dqp = doff_b + (long long)(3 * t) * p.P + ppos;
      const Corner cn = corner_at(p, pvalid, zb, yb, xbase, it, ocur);
      const Samp sp = make_samp_nb(p, g, c, cn);
      float* col = s_x + (i % 3) * XT + vox;
      float gd, gh, gw;
      if (half == 0) {
        rs_sample_half<CH, 0>(p, g, sp, cn, s_reg, xb, c0, chan, col, gd, gh, gw);
        pg[0] = gd; pg[1] = gh; pg[2] = gw;
      } else {
        rs_sample_half<CH, 1>(p, g, sp, cn, s_reg, xb, c0, chan, col, gd, gh, gw);
        float* pp = s_part + (i & 1) * 768 + vox;
        pp[0] = gd; pp[256] = gh; pp[512] = gw;
      }
      tap_next(p, it);
      DPF_STAMP(i, 1)
      __syncthreads();                                 // step barrier
      if (last_tap) {
        t = 0; c0 += CH; it = TapIt{0, 0, 0};
        if (i + 1 < NS) {
          stage_region_any<CH>(p, g, c, xb, c0, s_reg, tid, 1024, vec != 0);
          __syncthreads();
        }
      } else {
        ++t;
      }
    }
  } else if (role == 1) {
    // ------------------------------------------------------------------------------------------------ gcol = W^T . go
    // B fragments: go[k][voxel] for this wave's 4 sub-tiles of 16 voxels, all k (K <= 64)
    float bfrag[4][16];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int pl = rw * 64 + st * 16 + l15;
      const int ax = pl & (g.TX - 1), ay = (pl >> g.TXS) & (g.TY - 1), az = pl >> 6;
      const int gz = c.z0 + az, gy = c.y0 + ay, gx = c.x0 + ax;
      const bool ok = az < g.TZ && gz < p.Do && gy < p.Ho && gx < p.Wo;
      const long long gpos = ((long long)gz * p.Ho + gy) * p.Wo + gx;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int k = 4 * ks + lg;
        bfrag[st][ks] = (ok && k < p.K) ? go[((long long)c.b * p.K + k) * p.P + gpos] : 0.f;
      }
    }
    // A fragments of step j: W[k = 4 ks + lg][cj + l15][tj] (rows k >= K and columns >= C of the repacked tensor are zero), fetched one step ahead
    float aN[16];
    {
      const float* wtt = wt2 + (long long)lg * CT + l15;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) aN[ks] = wtt[(4 * ks) * CT];
    }
    int tj = 0, cj = 0;                                // (tap, chunk origin) of step j
#pragma unroll 1
    for (int j = 0; j <= NS; ++j) {                    // iteration j writes gcol(j) during step j - 1 (j = 0: the prologue)
      int tn = tj + 1, cn0 = cj;                       // step j + 1
      if (tn == p.T) { tn = 0; cn0 += CH; }
      DPF_STAMP(j, 0)
      if (j < NS) {
        float a[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) a[ks] = aN[ks];
        if (j + 1 < NS) {
          const float* wtt = wt2 + ((long long)tn * 64 + lg) * CT + cn0 + l15;
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) aN[ks] = wtt[(4 * ks) * CT];
        }
        float* dst = s_x + (j % 3) * XT + rw * 64 + l15;
#pragma unroll
        for (int sp2 = 0; sp2 < 2; ++sp2) {            // two sub-tiles at a time: 8 accumulator registers
          f32x4 acc[2];
          acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], bfrag[2 * sp2 + u][ks], acc[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (CH == 16 || 4 * lg + r < CH) dst[(4 * lg + r) * XS + (2 * sp2 + u) * 16] = acc[u][r];   // D row = channel, col = voxel
        }
      }
      DPF_STAMP(j, 1)
      __syncthreads();                                 // j = 0: prologue barrier; else the barrier of step j - 1
      tj = tn; cj = cn0;                               // now (tap, chunk) of step j + 1
      if (j >= 1 && j < NS && (j % p.T) == 0) {        // step j - 1 was the last tap of its chunk: help re-staging the region for step j
        stage_region_any<CH>(p, g, c, xb, (j / p.T) * CH, s_reg, tid, 1024, vec != 0);
        __syncthreads();
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------ dW partial = go . S^T
    // A fragments: go[k = 16 rw + l15][voxel], 64 k-steps over the 256 voxels of the tile.  The contraction order is free: k-step ks = 4 q + u
    // pairs matrix row lg with voxel 16 q + 4 lg + u, so the four B operands of steps 4 q .. 4 q + 3 are ONE ds_read_b128 per lane (64
    // single-dword reads per step left this role waiting on the LDS queue behind the samplers: 7.9 k of a 9.4 k-clock step busy)
    float wfrag[64];
    const int kk = 16 * rw + l15;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) {
      const int pl = 16 * (ks >> 2) + 4 * lg + (ks & 3);
      const int ax = pl & (g.TX - 1), ay = (pl >> g.TXS) & (g.TY - 1), az = pl >> 6;
      const int gz = c.z0 + az, gy = c.y0 + ay, gx = c.x0 + ax;
      const bool ok = kk < p.K && az < g.TZ && gz < p.Do && gy < p.Ho && gx < p.Wo;
      wfrag[ks] = ok ? go[((long long)c.b * p.K + kk) * p.P + ((long long)gz * p.Ho + gy) * p.Wo + gx] : 0.f;
    }
    float* rep = det ? dwtmp : dwtmp + (long long)(blockIdx.x % WG_NREP) * p.T * nchunk * 64 * 16;
    long long* rep_shadow = det ? reinterpret_cast<long long*>(dwtmp) : nullptr;       // deterministic mode: dcn_internal.h
    const int brow = (l15 < CH ? l15 : CH - 1) * XS + 4 * lg;  // B operand: S[channel l15][voxels 16 q + 4 lg ..+3]; rows beyond CH are masked below
    stage_region_any<CH>(p, g, c, xb, 0, s_reg, tid - 256, 768, vec != 0);
    __syncthreads();                                   // prologue barrier
    int ts = 0, cs = 0;                                // (tap, chunk index) of the step whose samples are contracted next
#pragma unroll 1
    for (int i = 0; i <= NS; ++i) {                    // iteration i contracts S(i - 1) during step i (i = NS: after the last barrier)
      DPF_STAMP(i, 0)
      if (i >= 1) {
        const float* src = s_x + ((i - 1) % 3) * XT + brow;
        f32x4 wacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
          const float4 bv = *reinterpret_cast<const float4*>(src + 16 * qq);
          wacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 0], bv.x, wacc[0], 0, 0, 0);
          wacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 1], bv.y, wacc[1], 0, 0, 0);
          wacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 2], bv.z, wacc[2], 0, 0, 0);
          wacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 3], bv.w, wacc[3], 0, 0, 0);
        }
        if (l15 < CH && cs * CH + l15 < p.C) {
          float* dst = rep + ((long long)(ts * nchunk + cs) * 64 + 16 * rw + 4 * lg) * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * rw + 4 * lg + r < p.K) dcn_acc_add(rep, rep_shadow, &dst[r * 16], (wacc[0][r] + wacc[1][r]) + (wacc[2][r] + wacc[3][r]));
        }
        if (++ts == p.T) { ts = 0; ++cs; }
      }
      DPF_STAMP(i, 1)
      if (i < NS) {
        __syncthreads();                               // barrier of step i
        if (i + 1 < NS && (i + 1) % p.T == 0) {        // step i was the last tap of its chunk
          stage_region_any<CH>(p, g, c, xb, ((i + 1) / p.T) * CH, s_reg, tid, 1024, vec != 0);
          __syncthreads();
        }
      }
    }
  }
}

// dW[k][c][t] = sum_rep tmp[rep][t][c/ch][k][c%ch]   (ch = channels per chunk of the producing kernel)
// det: the scratch holds ONE replica of order-independent integer pairs (dcn_internal.h)
__global__ void dcn_wgrad_fold_kernel(const float* __restrict__ dwtmp, float* __restrict__ dw, int K, int C, int T, int nchunk, int ch, int det) {
  const int total = K * C * T;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i % T, cc = (i / T) % C, k = i / (T * C);
    float s = 0.f;
    if (det) s = dpf_det_value(reinterpret_cast<const long long*>(dwtmp) + 2 * ((((long long)t) * nchunk + cc / ch) * 64 * 16 + k * 16 + (cc % ch)));
    else
      for (int r = 0; r < WG_NREP; ++r) s += dwtmp[(((long long)r * T + t) * nchunk + cc / ch) * 64 * 16 + k * 16 + (cc % ch)];
    dw[i] = s;
  }
}

// deterministic mode: grad_input = value of its integer shadow (every contribution went there; the tensor itself was only zero-filled)
__global__ void dcn_gi_finalize_kernel(const long long* __restrict__ shadow, float* __restrict__ gi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) gi[i] = dpf_det_value(shadow + 2 * i);
}

size_t region_lds(const RegGeo& g, int CH) { return sizeof(float) * ((size_t)CH * g.RV + (size_t)16 * ST); }

int region_geo(RegGeo& g, const DcnP& p, int CH, int R, bool aligned = false, int RYH = -1, int TX = RG_TX, int TY = RG_TY) {
  g.R = R;
  g.TX = TX;
  g.TY = TY;
  g.TXS = 0;
  while ((1 << g.TXS) < TX) ++g.TXS;
  g.RXL = R;
  g.RYH = RYH < R ? R : RYH;
  g.TZ = p.Do < 4 ? p.Do : 4;
  int RZ = (g.TZ - 1) * p.sd + (p.kd - 1) * p.dd + 1 + 2 * R;
  if (RZ > p.D) RZ = p.D;
  g.RZmax = RZ;
  g.RY = (TY - 1) * p.sh + (p.kh - 1) * p.dh + 1 + 2 * g.RYH;
  g.RX = (TX - 1) * p.sw + (p.kw - 1) * p.dw + 1 + 2 * R;
  if (aligned) {   // first region column = x0*sw - pw - RXL on a 16-byte boundary (x0*sw is a multiple of 32), RX a multiple of 4
    while ((p.pw + g.RXL) & 3) ++g.RXL;
    g.RX = ((g.RXL + (TX - 1) * p.sw + (p.kw - 1) * p.dw + 1 + R + 3) / 4) * 4;
  }
  g.RV = g.RZmax * g.RY * g.RX;
  g.tilesZ = dpf_div_up(p.Do, g.TZ);
  g.tilesY = dpf_div_up(p.Ho, TY);
  g.tilesX = dpf_div_up(p.Wo, TX);
  const long long blocks = (long long)p.B * g.tilesZ * g.tilesY * g.tilesX;
  if (g.RX > 64 || region_lds(g, CH) > 160 * 1024 || blocks >= 0x7fffffffLL || p.K > 128) return DPF_ERR_UNSUPPORTED;
  return DPF_OK;
}

// chunk width: 12 where it pads the channel count less than 16 does (35 -> 36 vs 48)
int region_chunk(int C) { return ((C + 11) / 12 * 12 < (C + 15) / 16 * 16) ? 12 : 16; }

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
// floats of the repacked-weights region at the head of the workspace: the largest of every kernel family's repack (region / gather kernels:
// reduce index padded; lean forward and lean backward fragment orders, which can exceed it -- C = 49..60 or 81..84 with 12-wide chunks).
// The grad_weight replicas start right behind it, so the SAME number sizes the workspace and places them (they overlapped).
static long long dcn_repack_floats(int C, int K, int T) {
  long long repack = (long long)T * (((C + 31) / 32) * 32) * (((K + 63) / 64) * 64);
  if (T == 27 && dcn_lean_workspace_floats(C, K) > repack) repack = dcn_lean_workspace_floats(C, K);
  return repack;
}

static long long dcn_workspace_floats(int C, int K, int T) {
  return (dcn_repack_floats(C, K, T) + (long long)WG_NREP * T * ((C + 11) / 12) * 64 * 16 + 64 + 3) / 4 * 4;   // + grad_weight scratch replicas (chunks of >= 12 channels) + max|W| per channel chunk
}
long long dpf_deform_conv3d_workspace_floats(int C, int K, int T) { return dcn_workspace_floats(C, K, T); }
// workspace of dpf_deform_conv3d_backward*: in deterministic mode (dpf_set_deterministic) grad_input is accumulated as order-independent
// integer pairs in a shadow behind the ordinary workspace -- 4 more floats per element of the input tensor
long long dpf_deform_conv3d_backward_workspace_floats(int B, int C, int D, int H, int W, int K, int T) {
  return dcn_workspace_floats(C, K, T) + (dpf_deterministic() ? 4LL * B * C * D * H * W : 0);
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
  const int CH = region_chunk(C);
  // (1) the configuration the model runs (3x3x3, stride 1, padding 1, dilation 1, depth <= 4, aligned rows): lean-sampler kernels of dcn_lean.hip
  if (kd == 3 && kh == 3 && kw == 3 && sd == 1 && sh == 1 && sw == 1 && pd == 1 && ph == 1 && pw == 1 && dd == 1 && dh == 1 && dw == 1 &&
      !getenv("DPF_DCN_V1")) {
    rc = dcn_lean_forward(input, offset, weight, bias, output, ws, B, C, D, H, W, K, st);
    if (rc != DPF_ERR_UNSUPPORTED) return rc;
  }
  // (2) any other geometry whose haloed LDS image fits: role-split region kernel (sampler waves + MFMA waves, double-buffered sample tile);
  // half-width tile (two workgroups per CU) for 16-channel chunks, full tile for 12-channel ones; with 16-byte aligned rows the region
  // gets an aligned x-origin and is staged with float4 loads
  if (!getenv("DPF_DCN_V1") && MT <= 2) {
    RegGeo ga{};
    const bool can_vec = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(input) % 16 == 0) && sw <= 2;
    const int TXv = CH == 16 ? 16 : 32;
    auto lds_of = [&](const RegGeo& q) { return sizeof(float) * ((size_t)CH * q.RV + (size_t)2 * CH * 8 * TXv); };
    const size_t lds_cap = TXv == 16 ? 80 * 1024 : 160 * 1024;
    int vec = 0;
    bool ok = false;
    if (can_vec) {
      const int cand[6][2] = {{4, 6}, {4, 5}, {4, 4}, {3, 5}, {3, 4}, {3, 3}};     // (x/z halo, y halo), widest first
      for (int i = 0; i < 6 && !ok; ++i)
        if (region_geo(ga, p, CH, cand[i][0], true, cand[i][1], TXv, 2) == DPF_OK && lds_of(ga) <= lds_cap) { ok = true; vec = 1; }
    }
    if (!ok) {
      for (int R = 4; R >= 3 && !ok; --R)
        if (region_geo(ga, p, CH, R, false, -1, TXv, 2) == DPF_OK && lds_of(ga) <= lds_cap) ok = true;
    }
    if (ok) {
      const int Cpad = (C + CH - 1) / CH * CH;
      hipLaunchKernelGGL(repack_weights_pad_kernel, dim3(dpf_ew_grid((long long)p.T * Cpad * KT)), dim3(256), 0, st, weight, ws, K, C, p.T, KT, 0,
                         Cpad);
      const size_t lds_rs = lds_of(ga);
      const dim3 grid_rs((unsigned)((long long)B * ga.tilesZ * ga.tilesY * ga.tilesX));
#define DPF_RS(M, Cw, Nw, Tx)                                                                                              \
  {                                                                                                                        \
    if (set_lds(dcn_fwd_rs_kernel<M, Cw, Nw, Tx>, lds_rs) != DPF_OK) return DPF_ERR_LAUNCH;                                \
    hipLaunchKernelGGL((dcn_fwd_rs_kernel<M, Cw, Nw, Tx>), grid_rs, dim3(64 * Nw), lds_rs, st, input, offset, ws, bias, output, p, ga, vec); \
  }
      if (MT == 1) { if (CH == 16) DPF_RS(1, 16, 8, 16) else DPF_RS(1, 12, 8, 32) } else { if (CH == 16) DPF_RS(2, 16, 8, 16) else DPF_RS(2, 12, 8, 32) }
#undef DPF_RS
      return dpf_check_launch();
    }
  }
  // (3) everything else: one workgroup per 64 output voxels, samples gathered from global memory
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
int dpf_deform_conv3d_backward_ex(const float* input, const float* weight, const float* bias, const float* offset, const float* grad_output,
                                  float* grad_input, float* grad_offset, float* grad_weight, float* grad_bias, float* ws, int B, int C,
                                  int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
                                  int dh, int dw, int group, int deformable_group, int im2col_step, int grad_input_channels, void* stream);

int dpf_deform_conv3d_backward(const float* input, const float* weight, const float* bias, const float* offset, const float* grad_output,
                               float* grad_input, float* grad_offset, float* grad_weight, float* grad_bias, float* ws, int B, int C,
                               int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
                               int dh, int dw, int group, int deformable_group, int im2col_step, void* stream) {
  return dpf_deform_conv3d_backward_ex(input, weight, bias, offset, grad_output, grad_input, grad_offset, grad_weight, grad_bias, ws, B, C, D, H,
                                       W, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw, group, deformable_group, im2col_step, C, stream);
}

// Same as dpf_deform_conv3d_backward, but grad_input is only produced for input channels [0, grad_input_channels) (the remaining
// channels of the zero-initialised tensor stay 0): StereoDPNet's first deformable conv consumes 32 cost channels + 3 constant
// XYZ channels (normal_module.py:166), whose gradient nobody reads.
int dpf_deform_conv3d_backward_ex(const float* input, const float* weight, const float* bias, const float* offset, const float* grad_output,
                                  float* grad_input, float* grad_offset, float* grad_weight, float* grad_bias, float* ws, int B, int C,
                                  int D, int H, int W, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
                                  int dh, int dw, int group, int deformable_group, int im2col_step, int grad_input_channels, void* stream) {
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
  // deterministic mode: every merge of partial results is order-independent (dcn_internal.h) -- grad_input through an integer shadow behind
  // the workspace (the caller sized it with dpf_deform_conv3d_backward_workspace_floats), the grad_weight scratch as one integer replica
  const int det = dpf_deterministic();
  long long* gi_shadow = nullptr;
  if (det) {
    gi_shadow = reinterpret_cast<long long*>(ws + dcn_workspace_floats(C, K, p.T));
    if (reinterpret_cast<uintptr_t>(gi_shadow) & 7) return DPF_ERR_INVALID_ARG;
    if (hipMemsetAsync(gi_shadow, 0, sizeof(long long) * 2 * (size_t)in_elems, st) != hipSuccess) return DPF_ERR_LAUNCH;
  }
  if (hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)K * C * p.T, st) != hipSuccess) return DPF_ERR_LAUNCH;
  // wt2[T][K][CT]: reduce = K (A), out = C (B); the region kernels read it with K zero-padded to 64 rows
  // grad_input: LDS-privatised scatter when the haloed region fits, else the reference-style global atomics
  bool dx_done = false;
  if (K <= 64) {
    GiP q{};
    q.TZ = p.Do < 4 ? p.Do : 4;
    int RZ = (q.TZ - 1) * sd + (kd - 1) * dd + 1 + 2 * GI_R;
    if (RZ > D) RZ = D;
    q.RZmax = RZ;
    q.RY = (GI_TY - 1) * sh + (kh - 1) * dh + 1 + 2 * GI_R;
    q.RX = (GI_TX - 1) * sw + (kw - 1) * dw + 1 + 2 * GI_R;
    q.tilesZ = dpf_div_up(p.Do, q.TZ);
    q.tilesY = dpf_div_up(p.Ho, GI_TY);
    q.tilesX = dpf_div_up(p.Wo, GI_TX);
    q.CG = grad_input_channels < C ? (grad_input_channels < 0 ? 0 : grad_input_channels) : C;
    const int npos = 64 * q.TZ;
    // region (8 packed pairs per cell + dummy cell), 2 x (lidx, w) tables, far flags, mass counters
    const size_t lds = sizeof(long long) * (size_t)(q.RZmax * q.RY * q.RX + 1) * PK_CS + sizeof(float) * ((size_t)npos * 36 + 4 + 48) +
                       sizeof(unsigned) * (size_t)((q.RZmax * q.RY * q.RX + 4) & ~3);
    const long long blocks = (long long)B * q.tilesZ * q.tilesY * q.tilesX;
    if (lds <= 160 * 1024 && blocks < 0x7fffffffLL && (long long)D * H * W < 0x7fffffffLL) {
      const dim3 grid((unsigned)blocks);
      // packed fixed-point region (two channels per ds_add_u64): per-chunk max |W| for the quantisation bound, kept behind the
      // grad_weight scratch in ws (the last entry of that 64-float slack: the weight exponent of the f16-component path)
      float* wmaxv = ws + dcn_repack_floats(C, K, p.T) + (long long)WG_NREP * p.T * ((C + 11) / 12) * 64 * 16;
      int* wexp = reinterpret_cast<int*>(wmaxv + 63);
      static const int gh_env = getenv("DPF_DCN_GCOL16") ? atoi(getenv("DPF_DCN_GCOL16")) : 1;
      // (matrix path 1 is exact per element everywhere: the f16 components serve path 2 only; the per-chunk maxima must leave the exponent's slot alone)
      const bool f16 = gh_env && dpf_conv_f32_x9() == 2 && p.T == 27 && dpf_div_up(C, GI_CH) <= 63;
      if (f16) {
        // the gcol B operand as f16 fragments ([T][CT / 16][4 KB] -- the bytes of wt2[T][64][CT]), max |W| from the caller's tensor
        hipLaunchKernelGGL(dcn_repack_pk_h_kernel, dim3(16), dim3(1024), 0, st, weight, reinterpret_cast<unsigned short*>(ws), K, C, p.T, CT / PK_CH, wexp);
        hipLaunchKernelGGL(dcn_wmax_w_kernel, dim3(dpf_div_up(C, GI_CH)), dim3(256), 0, st, weight, wmaxv, K, C, p.T);
      } else {
        hipLaunchKernelGGL(repack_weights_pad_kernel, dim3(dpf_ew_grid((long long)p.T * 64 * CT)), dim3(256), 0, st, weight, ws, K, C, p.T, CT, 1,
                           64);
        hipLaunchKernelGGL(dcn_wmax_kernel, dim3(dpf_div_up(C, GI_CH)), dim3(256), 0, st, ws, wmaxv, p.T, CT, C);
      }
#define DPF_GIP2(NS, NWv, F)                                                                                                   \
  {                                                                                                                            \
    if (set_lds(dcn_bwd_input_pk_kernel<NS, NWv, F>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                                    \
    hipLaunchKernelGGL((dcn_bwd_input_pk_kernel<NS, NWv, F>), grid, dim3(64 * NWv), lds, st, offset, ws, grad_output, grad_input, p, q, CT, wmaxv, gi_shadow, wexp); \
  }
#define DPF_GIP(NS, NWv) { if (f16) DPF_GIP2(NS, NWv, true) else DPF_GIP2(NS, NWv, false) }
      switch (q.TZ) {
        case 1: DPF_GIP(1, 4); break;
        case 2: DPF_GIP(2, 4); break;
        case 3: DPF_GIP(3, 4); break;
        default: DPF_GIP(1, 16); break;   // 256 voxels: 16 waves (four per SIMD, one 16-voxel sub-tile each; the 8-wave variant measured 9.6 vs 8.5 ms)
      }
#undef DPF_GIP
#undef DPF_GIP2
      dx_done = true;
    }
  }
  // grad_offset + grad_weight.  (1) the model's configuration: lean-sampler kernel of dcn_lean.hip; (2) any other geometry whose haloed
  // LDS image fits next to three rotating [CH][256] tiles: role-split region kernel (sampler / gcol / wgrad waves); both write grad_weight
  // partials into dwtmp[8][T][nchunk][64][16], folded below; (3) everything else: global-memory gather kernels
  const int CHb = region_chunk(C);
  const bool can_vec = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(input) % 16 == 0) && sw <= 2;
  const bool region_ok = dx_done && K <= 64 && !getenv("DPF_DCN_V1") && (CHb == 16 || (C + 11) / 12 * 12 + 4 <= CT);   // 16 weight columns are fetched from each chunk origin
  float* dwtmp = ws + dcn_repack_floats(C, K, p.T);
  const int nchunk = (C + CHb - 1) / CHb;
  bool rs_done = false;
  if (region_ok) {
    if (hipMemsetAsync(dwtmp, 0, sizeof(float) * (size_t)WG_NREP * p.T * nchunk * 64 * 16, st) != hipSuccess) return DPF_ERR_LAUNCH;
    if (kd == 3 && kh == 3 && kw == 3 && sd == 1 && sh == 1 && sw == 1 && pd == 1 && ph == 1 && pw == 1 && dd == 1 && dh == 1 && dw == 1 &&
        CHb == dcn_lean_chunk(C)) {
      rc = dcn_lean_bwd_offset(input, offset, weight, grad_output, grad_offset, dwtmp, ws, B, C, D, H, W, K, st, det);
      if (rc == DPF_OK) rs_done = true;
      else if (rc != DPF_ERR_UNSUPPORTED) return rc;
    }
  }
  if (!rs_done && region_ok) {
    RegGeo gr{};
    auto lds_of = [&](const RegGeo& qq) { return sizeof(float) * ((size_t)CHb * qq.RV + (size_t)3 * CHb * XS + 2 * 3 * 256); };
    bool ok = false;
    int vec_rs = 0;
    if (can_vec) {
      // compact 4 x 4 x 16 tile: the 256 voxels need a smaller haloed region than 4 x 2 x 32, so a wider halo fits next to the three
      // rotating tiles -- less staging per tile and fewer samples on the global slow path
      const int cand[7][2] = {{5, 6}, {4, 6}, {4, 5}, {4, 4}, {3, 5}, {3, 4}, {3, 3}};
      for (int i = 0; i < 7 && !ok; ++i)
        if (region_geo(gr, p, CHb, cand[i][0], true, cand[i][1], 16, 4) == DPF_OK && lds_of(gr) <= 160 * 1024) { ok = true; vec_rs = 1; }
    }
    if (!ok) {
      for (int R = 4; R >= 3 && !ok; --R)
        if (region_geo(gr, p, CHb, R) == DPF_OK && lds_of(gr) <= 160 * 1024) ok = true;
    }
    if (ok) {
      // (the lean kernel may have overwritten ws: wt2[T][64][CT] again)
      hipLaunchKernelGGL(repack_weights_pad_kernel, dim3(dpf_ew_grid((long long)p.T * 64 * CT)), dim3(256), 0, st, weight, ws, K, C, p.T, CT, 1, 64);
      const size_t lds = lds_of(gr);
      const dim3 grid((unsigned)((long long)B * gr.tilesZ * gr.tilesY * gr.tilesX));
#define DPF_OFFRS(Cw)                                                                                                          \
  {                                                                                                                            \
    if (set_lds(dcn_bwd_offset_rs_kernel<Cw>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                                           \
    hipLaunchKernelGGL((dcn_bwd_offset_rs_kernel<Cw>), grid, dim3(1024), lds, st, input, offset, ws, grad_output, grad_offset, dwtmp, p, gr, \
                       CT, nchunk, vec_rs, det);                                                                               \
  }
      if (CHb == 16) DPF_OFFRS(16) else DPF_OFFRS(12)
#undef DPF_OFFRS
      rs_done = true;
    }
  }
  if (rs_done) {
    hipLaunchKernelGGL(dcn_wgrad_fold_kernel, dim3(dpf_ew_grid((long long)K * C * p.T)), dim3(256), 0, st, dwtmp, grad_weight, K, C, p.T, nchunk, CHb, det);
  } else {
    hipLaunchKernelGGL(repack_weights_kernel, dim3(dpf_ew_grid((long long)p.T * K * CT)), dim3(256), 0, st, weight, ws, K, C, p.T, CT, 1);
    {
      const size_t lds = sizeof(float) * ((size_t)K * SP + (size_t)CT * SP + 3 * 4 * TP);
      const dim3 grid((unsigned)(B * p.tiles_per_b));
#define DPF_D(M)                                                                                                       \
  {                                                                                                                    \
    if (dx_done) {                                                                                                     \
      if (set_lds(dcn_bwd_data_kernel<M, false>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                                \
      hipLaunchKernelGGL((dcn_bwd_data_kernel<M, false>), grid, dim3(256), lds, st, input, offset, ws, grad_output, grad_input, grad_offset, p, gi_shadow); \
    } else {                                                                                                           \
      if (set_lds(dcn_bwd_data_kernel<M, true>, lds) != DPF_OK) return DPF_ERR_LAUNCH;                                 \
      hipLaunchKernelGGL((dcn_bwd_data_kernel<M, true>), grid, dim3(256), lds, st, input, offset, ws, grad_output, grad_input, grad_offset, p, gi_shadow); \
    }                                                                                                                  \
  }
      switch (MTC) { case 1: DPF_D(1); break; case 2: DPF_D(2); break; case 3: DPF_D(3); break; default: DPF_D(4); break; }
#undef DPF_D
    }
    const long long ntile = (long long)B * p.tiles_per_b;
    long long nchunkw = 2048 / p.T;
    if (nchunkw < 1) nchunkw = 1;
    if (nchunkw > ntile) nchunkw = ntile;
    if (det) nchunkw = 1;            // each dW address then receives ONE atomic add: order-independent
    p.nchunk = (int)nchunkw;
    // LDS of the INSTANTIATION that runs (<4, 4> for more than two row tiles on either side lays its tiles out for 4 + 4: sizing it for
    // the actual counts put the grad_output tile outside the allocation -- C = 84 on this path returned a zero grad_weight)
    const bool small = MT <= 2 && MTC <= 2;
    const size_t lds = sizeof(float) * ((size_t)32 * (small ? MTC : 4) * SP + (size_t)32 * (small ? MT : 4) * SP);
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
  if (det) hipLaunchKernelGGL(dcn_gi_finalize_kernel, dim3(dpf_ew_grid(in_elems)), dim3(256), 0, st, gi_shadow, grad_input, in_elems);
  if (grad_bias) {
    // grad_bias[k] = sum_{b,p} go[b,k,p]  (cu:277) -- small row reduction
    if (hipMemsetAsync(grad_bias, 0, sizeof(float) * K, st) != hipSuccess) return DPF_ERR_LAUNCH;
    rc = dpf_channel_sum(grad_output, grad_bias, B, K, p.P, stream);
    if (rc != DPF_OK) return rc;
  }
  return dpf_check_launch();
}

#ifdef DPF_STAMPS
int dpf_debug_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16 * 128 * 2) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
