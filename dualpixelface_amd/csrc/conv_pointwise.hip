// Pointwise (1x1 / 1x1x1) convolutions: conv5.pointwise / conv_skip of the DPBlocks, the FPN lateral convs, mask_convs.3.0 of the
// attention layer, lastconv.2 / downsample / SPP branch convs of the PSMNet-style extractors
// (reference: src/module/asm/basics.py:39-58, src/model/stereodpnet/modules.py:26-32,83-85, src/module/asm/asm.py:141-146).
//
// These are HBM-bound (64-256 FLOP per 8 bytes moved): the job is to read every input element once with 16-byte loads, keep the
// small weight matrix in LDS and write 16-byte stores.  The tiled implicit-GEMM kernels stage a haloed patch per tile, which for a
// 1x1 window is pure overhead (1.4-1.8 TB/s); here a wave owns 128 consecutive positions:
//   forward / data gradient: v_mfma_f32_32x32x2_f32 with the weight matrix as A (row = output channel) and, as B, the float4 a
//     lane loads from one input channel -- column j of position tile t IS position 4 j + t, so the four B operands of a lane are
//     the four components of its load and its 16 accumulator rows leave as float4 stores.  No LDS traffic for activations at all.
//   weight gradient: dW[k][c] = sum_p g[k][p] x[c][p]; g and x tiles go through LDS (coalesced float4 rows in, transposed
//     fragments out, stride PT + 2 floats = conflict-free), per-workgroup partial matrices in a slab, fixed-order reduce.
// Stride 2 (conv_skip of the down-sampling DPBlocks) is handled for exact halving (IH = 2 OH, IW = 2 OW).
#include "conv_internal.h"

namespace {

struct PwP {
  int N, C, K, Ktot, k0;        // C = reduce channels, K = output channels of this launch (<= 64 per slice)
  long long Pin, Pout;          // positions per sample of the input / output tensor
  int OW, IW;                   // row lengths (stride-2 modes)
  int mode;                     // weight indexing: 0 = w[o * wB + r], 1 = w[r * wB + o]
  int wB;
  int stride;                   // 1; 2 = forward reads every other pixel / row; -2 = transposed: writes every other pixel / row
  long long in_plane_rows;      // OH (stride 2 bookkeeping)
};

template <int MT>
__global__ __launch_bounds__(256) void pointwise_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ out, PwP p) {
  extern __shared__ float wl[];                 // [C (padded to even)][KT]
  constexpr int KT = 32 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int ks = blockIdx.z * KT;               // first output channel of this slice (relative to k0)
  const int Cp = (p.C + 1) & ~1;
  for (int i = tid; i < Cp * KT; i += 256) {
    const int r = i / KT, o = i - r * KT;
    float v = 0.f;
    if (r < p.C && ks + o < p.K) v = p.mode == 0 ? w[(long long)(p.k0 + ks + o) * p.wB + r] : w[(long long)r * p.wB + p.k0 + ks + o];
    wl[i] = v;
  }
  __syncthreads();
  const int n = blockIdx.y;
  const long long pos = (long long)blockIdx.x * 512 + wave * 128 + 4 * l31;     // first of this lane's 4 (output-grid) positions
  const bool live = pos < (p.stride == 2 ? p.Pout : (p.stride == -2 ? p.Pin : p.Pout));
  const float* xn = x + (long long)n * p.C * p.Pin;
  long long ioff = pos;                          // offset of the lane's first input element inside a channel plane
  if (p.stride == 2) {
    const long long oy = pos / p.OW, ox = pos - oy * p.OW;
    ioff = (2 * oy) * p.IW + 2 * ox;
  }
  f32x16 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;
  const int ncp = Cp >> 1;
#pragma unroll 4
  for (int cp = 0; cp < ncp; ++cp) {
    const int c = 2 * cp + hi;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && c < p.C) {
      const float* src = xn + (long long)c * p.Pin + ioff;
      if (p.stride == 2) {
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        v = make_float4(a.x, a.z, b.x, b.z);
      } else {
        v = *reinterpret_cast<const float4*>(src);
      }
    }
    const float* wrow = wl + c * KT + l31;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float a = wrow[m * 32];
      acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.x, acc[m][0], 0, 0, 0);
      acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.y, acc[m][1], 0, 0, 0);
      acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.z, acc[m][2], 0, 0, 0);
      acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.w, acc[m][3], 0, 0, 0);
    }
  }
  if (!live) return;
  float* on = out + ((long long)n * p.Ktot + p.k0 + ks) * p.Pout;
  long long ooff = pos;
  if (p.stride == -2) {                         // transposed stride 2: q-grid position -> even row / even columns of the dense grid
    const long long qy = pos / p.IW, qx = pos - qy * p.IW;          // here IW = row length of the small (input) grid
    ooff = (2 * qy) * p.OW + 2 * qx;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hi;
      if (ks + k < p.K) {
        const float bv = bias ? bias[p.k0 + ks + k] : 0.f;
        float* o = on + (long long)k * p.Pout + ooff;
        const float4 r = make_float4(acc[m][0][j] + bv, acc[m][1][j] + bv, acc[m][2][j] + bv, acc[m][3][j] + bv);
        if (p.stride == -2) {
          *reinterpret_cast<float4*>(o) = make_float4(r.x, bv, r.y, bv);
          *reinterpret_cast<float4*>(o + 4) = make_float4(r.z, bv, r.w, bv);
          *reinterpret_cast<float4*>(o + p.OW) = make_float4(bv, bv, bv, bv);
          *reinterpret_cast<float4*>(o + p.OW + 4) = make_float4(bv, bv, bv, bv);
        } else {
          *reinterpret_cast<float4*>(o) = r;
        }
      }
    }
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
constexpr int PT = 64;                 // positions per LDS tile
constexpr int LS = PT + 2;             // row stride (floats): 32 consecutive rows hit 32 distinct even banks, the odd k-half the odd ones

struct PwgP {
  int N, C, K, Ktot, k0;
  long long Pg, Px;                    // positions per sample of g (small grid) and x
  int QW, IW, stride;                  // stride 2: x is read at (2 qy, 2 qx)
  int chunk;                           // g positions per workgroup
  int tiles;                           // MT * CT output tiles
  int MT, CT;
  int G, S;                            // wave w: tile group w % G, position slice w / G
};

__global__ __launch_bounds__(256) void pointwise_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ slab,
                                                              PwgP p) {
  extern __shared__ float sm[];
  const int KT = 32 * p.MT, CTT = 32 * p.CT;
  float* gl = sm;                       // [KT][LS]
  float* xl = sm + KT * LS;             // [CTT][LS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const long long chunks_per_n = (p.Pg + p.chunk - 1) / p.chunk;
  const int n = (int)(blockIdx.x / chunks_per_n);
  const long long p0 = (blockIdx.x - (long long)n * chunks_per_n) * p.chunk;
  const long long pend = p0 + p.chunk < p.Pg ? p0 + p.chunk : p.Pg;
  const float* gn = g + ((long long)n * p.Ktot + p.k0) * p.Pg;
  const float* xn = x + (long long)n * p.C * p.Px;
  const int tg = wave % p.G, slice = wave / p.G;
  const int per = (PT / 2) / p.S;                        // MFMA k-steps of this wave's slice per tile
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;
  for (long long pt = p0; pt < pend; pt += PT) {
    __syncthreads();                                     // the previous tile has been consumed
    for (int i = tid; i < (KT + CTT) * (PT / 4); i += 256) {
      const int row = i / (PT / 4), q = (i - row * (PT / 4)) * 4;
      const long long pos = pt + q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      float* dst;
      if (row < KT) {
        dst = gl + row * LS + q;
        if (row < p.K && pos < pend) v = *reinterpret_cast<const float4*>(gn + (long long)row * p.Pg + pos);
      } else {
        const int c = row - KT;
        dst = xl + c * LS + q;
        if (c < p.C && pos < pend) {
          if (p.stride == 2) {
            const long long qy = pos / p.QW, qx = pos - qy * p.QW;
            const float* src = xn + (long long)c * p.Px + (2 * qy) * p.IW + 2 * qx;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v = make_float4(a.x, a.z, b.x, b.z);
          } else {
            v = *reinterpret_cast<const float4*>(xn + (long long)c * p.Px + pos);
          }
        }
      }
      reinterpret_cast<float2*>(dst)[0] = make_float2(v.x, v.y);    // LS is even, q % 4 == 0: 8-byte aligned, not 16
      reinterpret_cast<float2*>(dst)[1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    for (int i = slice * per; i < (slice + 1) * per; ++i) {
      const int kk = 2 * i + hi;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int t = tg + a * p.G;
        if (t < p.tiles) {
          const int m = t / p.CT, ct = t - m * p.CT;
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(gl[(m * 32 + l31) * LS + kk], xl[(ct * 32 + l31) * LS + kk], acc[a], 0, 0, 0);
        }
      }
    }
  }
  // one slab row per WORKGROUP: [K][C].  With S > 1 position slices (few output tiles: e.g. 32 x 32 -> 4 slices of one tile) the slices'
  // partial tiles are summed through LDS in a fixed order first (S = 2 or 4, <= 4 tiles: <= 16 KB), so that a row costs a workgroup, not a
  // wave: 4x fewer slab rows to write and fold (32 -> 32 at 4 x 256 x 384: 81 -> 51 us per call, 1.2 -> 2.0 TB/s)
  float* row = slab + (long long)blockIdx.x * p.K * p.C;
  if (p.S > 1) {
    __syncthreads();                                     // the last tile has been consumed: the staging area is free
    float* red = sm;                                     // [S][tiles][32 x 32]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int t = tg + a * p.G;
      if (t >= p.tiles) continue;
#pragma unroll
      for (int j = 0; j < 16; ++j) red[((slice * p.tiles + t) * 16 + j) * 64 + lane] = acc[a][j];
    }
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int t = tg + a * p.G;
      if (t >= p.tiles) continue;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = acc[a][j];
        for (int sl = 1; sl < p.S; ++sl) v += red[((sl * p.tiles + t) * 16 + j) * 64 + lane];
        acc[a][j] = v;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int t = tg + a * p.G;
    if (t >= p.tiles) continue;
    const int m = t / p.CT, ct = t - m * p.CT;
    const int c = ct * 32 + l31;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hi;
      if (k < p.K && c < p.C) row[(long long)k * p.C + c] = acc[a][j];
    }
  }
}

// two-level fixed-order fold of the slab rows: level 1 sums groups of RGROUP rows (grid.y = groups), level 2 the group sums
constexpr int RGROUP = 32;
__global__ void pointwise_wgrad_fold_kernel(const float* __restrict__ slab, float* __restrict__ part, long long n, int rows) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r0 = blockIdx.y * RGROUP, r1 = r0 + RGROUP < rows ? r0 + RGROUP : rows;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += slab[(long long)r * n + i];
  part[(long long)blockIdx.y * n + i] = s;
}
__global__ void pointwise_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long n, int groups, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? dw[i] : 0.f;
  for (int r = 0; r < groups; ++r) s += part[(long long)r * n + i];
  dw[i] = s;
}

int env_flag(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

// forward / transposed pointwise convolution; DPF_ERR_UNSUPPORTED -> the caller's other kernels take it
int dpf_pointwise_conv(const float* x, const float* w, const float* bias, float* out, const DpfConvDesc& d, hipStream_t st) {
  static const int enabled = env_flag("DPF_POINTWISE", 1);
  if (!enabled || d.kd * d.kh * d.kw != 1 || d.pd || d.ph || d.pw) return DPF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return DPF_ERR_UNSUPPORTED;
  PwP p{};
  p.N = d.N; p.C = d.C; p.K = d.K; p.Ktot = d.Ktot; p.k0 = d.k0;
  p.mode = d.mode; p.wB = d.wB;
  p.Pin = (long long)d.ID * d.IH * d.IW;
  p.Pout = (long long)d.OD * d.OH * d.OW;
  p.OW = d.OW; p.IW = d.IW;
  long long grid_pos;
  if (d.sd == 1 && d.sh == 1 && d.sw == 1) {
    if (p.Pin != p.Pout || (p.Pin & 3)) return DPF_ERR_UNSUPPORTED;
    p.stride = 1;
    grid_pos = p.Pout;
  } else if (d.sd == 1 && d.sh == 2 && d.sw == 2 && !d.transposed) {
    if (d.ID != 1 || d.IH != 2 * d.OH || d.IW != 2 * d.OW || (d.OW & 3)) return DPF_ERR_UNSUPPORTED;
    p.stride = 2;
    grid_pos = p.Pout;
  } else if (d.sd == 1 && d.sh == 2 && d.sw == 2 && d.transposed) {
    // x lives on the small grid [ID=1, IH, IW], out on the dense grid [OH = 2 IH, OW = 2 IW]
    if (d.ID != 1 || d.OH != 2 * d.IH || d.OW != 2 * d.IW || (d.IW & 3)) return DPF_ERR_UNSUPPORTED;
    p.stride = -2;
    grid_pos = p.Pin;
  } else {
    return DPF_ERR_UNSUPPORTED;
  }
  const int slices64 = (d.K + 63) / 64;
  const int MT = d.K <= 32 ? 1 : 2;
  const int KT = 32 * MT;
  const size_t lds = sizeof(float) * (size_t)((d.C + 1) & ~1) * KT;
  if (lds > 96 * 1024 || d.N > 65535) return DPF_ERR_UNSUPPORTED;
  const long long bx = (grid_pos + 511) / 512;
  if (bx > 0x7fffffffLL) return DPF_ERR_UNSUPPORTED;
  dim3 grid((unsigned)bx, (unsigned)d.N, (unsigned)(MT == 1 ? 1 : slices64));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pointwise_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pointwise_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  if (MT == 1)
    hipLaunchKernelGGL(pointwise_kernel<1>, grid, dim3(256), lds, st, x, w, bias, out, p);
  else
    hipLaunchKernelGGL(pointwise_kernel<2>, grid, dim3(256), lds, st, x, w, bias, out, p);
  return dpf_check_launch();
}

long long dpf_pointwise_wgrad_workspace_floats(int C, int K) { return 2048LL * K * C + 1024; }   // slab rows are capped by the workspace

int dpf_pointwise_wgrad(const float* g, const float* x, float* dw, float* ws, long long ws_floats, const DpfWgradDesc& d, int accumulate,
                        hipStream_t st) {
  static const int enabled = env_flag("DPF_POINTWISE", 1);
  if (!enabled || !ws || d.kd * d.kh * d.kw != 1 || d.pd || d.ph || d.pw) return DPF_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(g) & 15)) return DPF_ERR_UNSUPPORTED;
  PwgP p{};
  p.N = d.N; p.C = d.C; p.K = d.K; p.Ktot = d.Ktot; p.k0 = d.k0;
  p.Pg = (long long)d.QD * d.QH * d.QW;
  p.Px = (long long)d.ID * d.IH * d.IW;
  p.QW = d.QW; p.IW = d.IW;
  if (d.sd == 1 && d.sh == 1 && d.sw == 1) {
    if (p.Pg != p.Px) return DPF_ERR_UNSUPPORTED;
    p.stride = 1;
  } else if (d.sd == 1 && d.sh == 2 && d.sw == 2) {
    if (d.ID != 1 || d.IH != 2 * d.QH || d.IW != 2 * d.QW || (d.QW & 3)) return DPF_ERR_UNSUPPORTED;
    p.stride = 2;
  } else {
    return DPF_ERR_UNSUPPORTED;
  }
  if (p.Pg & 3) return DPF_ERR_UNSUPPORTED;
  p.MT = (d.K + 31) / 32;
  p.CT = (d.C + 31) / 32;
  p.tiles = p.MT * p.CT;
  if (p.tiles > 16) return DPF_ERR_UNSUPPORTED;
  p.G = p.tiles < 4 ? p.tiles : 4;
  p.S = 4 / p.G;
  if (p.G == 3) p.S = 1;                                     // 3 tile groups: the fourth wave idles
  // chunk: enough workgroups to fill the chip, rows of the slab bounded by the workspace
  long long chunk = 1024;
  const long long total = (long long)d.N * p.Pg;
  static const long long fill = 512;    // workgroups to aim for (2048 measured slower: more slab rows to fold)
  while (chunk > PT && total / chunk < fill) chunk /= 2;
  // workspace: slab rows (one per workgroup) + their group sums (rows / RGROUP + 1 more rows)
  auto need = [&](long long ch) {
    const long long rows_ = ((p.Pg + ch - 1) / ch) * d.N;
    return (rows_ + rows_ / RGROUP + 2) * (long long)d.K * d.C;
  };
  while (need(chunk) > ws_floats && chunk < (1LL << 24)) chunk *= 2;
  p.chunk = (int)chunk;
  const long long chunks_per_n = (p.Pg + chunk - 1) / chunk;
  const long long blocks = chunks_per_n * d.N;
  const long long rows = blocks;
  if (need(chunk) > ws_floats || blocks > 0x7fffffffLL || rows > 65535LL * RGROUP) return DPF_ERR_UNSUPPORTED;
  size_t lds = sizeof(float) * (size_t)(32 * p.MT + 32 * p.CT) * LS;
  if (p.S > 1 && lds < sizeof(float) * (size_t)p.S * p.tiles * 1024) lds = sizeof(float) * (size_t)p.S * p.tiles * 1024;   // slice reduction
  if (lds > 96 * 1024) return DPF_ERR_UNSUPPORTED;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pointwise_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  if (p.G == 3) {                                            // slice index = wave / 3: waves 0..2 slice 0, wave 3 would be slice 1
    p.G = 4;                                                 // use 4 groups with the last one possibly empty instead
    p.S = 1;
  }
  hipLaunchKernelGGL(pointwise_wgrad_kernel, dim3((unsigned)blocks), dim3(256), lds, st, g, x, ws, p);
  const long long n = (long long)d.K * d.C;
  const int groups = (int)((rows + RGROUP - 1) / RGROUP);
  float* part = ws + rows * n;
  hipLaunchKernelGGL(pointwise_wgrad_fold_kernel, dim3(dpf_div_up(n, 256), groups), dim3(256), 0, st, ws, part, n, (int)rows);
  hipLaunchKernelGGL(pointwise_wgrad_reduce_kernel, dim3(dpf_div_up(n, 256)), dim3(256), 0, st, part, dw, n, groups, accumulate);
  return dpf_check_launch();
}
