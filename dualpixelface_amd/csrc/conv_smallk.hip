// Convolutions with a handful of output channels (K <= 4): the 32 -> 1 cost heads of the aggregation stack
// (reference: src/model/stereodpnet/modules.py:286-296 `classif*[2]`) and the last 32 -> 3 normal conv
// (normal_module.py:65).  A 32-row MFMA tile would be 3-10 % utilised here; the work is HBM-bound instead
// (read C channels once, write K), so these are direct kernels: forward = one thread per output voxel with the weights
// broadcast from LDS; weight gradient = one thread per (channel, tap) pair marching over an LDS-staged voxel tile.
#include "dpf_common.h"

namespace {

constexpr int MAXK = 4;
constexpr int TW = 32, TH = 8;   // wgrad position tile

struct SkP {
  int N, C, K;
  int ID, IH, IW, OD, OH, OW;
  int kd, kh, kw, T;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
};

__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, SkP p) {
  extern __shared__ float s_w[];   // [K][C][T]
  for (int i = threadIdx.x; i < p.K * p.C * p.T; i += 256) s_w[i] = w[i];
  __syncthreads();
  const long long oplane = (long long)p.OH * p.OW, ovol = oplane * p.OD;
  const long long ivol = (long long)p.ID * p.IH * p.IW;
  const long long total = (long long)p.N * ovol;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow = (int)(i % p.OW);
    const int oh = (int)((i / p.OW) % p.OH);
    const int od = (int)((i / oplane) % p.OD);
    const int n = (int)(i / ovol);
    float acc[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[k] = (bias && k < p.K) ? bias[k] : 0.f;
    const float* xn = x + (long long)n * p.C * ivol;
    for (int a = 0; a < p.kd; ++a) {
      const int id = od * p.sd - p.pd + a * p.dd;
      if (id < 0 || id >= p.ID) continue;
      for (int b = 0; b < p.kh; ++b) {
        const int ih = oh * p.sh - p.ph + b * p.dh;
        if (ih < 0 || ih >= p.IH) continue;
        for (int c2 = 0; c2 < p.kw; ++c2) {
          const int iw = ow * p.sw - p.pw + c2 * p.dw;
          if (iw < 0 || iw >= p.IW) continue;
          const int t = (a * p.kh + b) * p.kw + c2;
          const float* xp = xn + ((long long)id * p.IH + ih) * p.IW + iw;
          for (int c = 0; c < p.C; ++c) {
            const float v = xp[(long long)c * ivol];
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
              if (k < p.K) acc[k] = fmaf(s_w[(k * p.C + c) * p.T + t], v, acc[k]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < p.K) out[((long long)n * p.K + k) * ovol + (i % ovol)] = acc[k];
  }
}

// grid = cchunks * nblk; block: channels [c0, c0+CCH), tiles pchunk, pchunk+nblk, ...; thread = (cc, tap) pair
template <int CCH>
__global__ __launch_bounds__(256) void smallk_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw, SkP p,
                                                           int nblk, int tilesH, int tilesW, long long ntiles) {
  extern __shared__ float smem[];
  const int ext_d = (p.kd - 1) * p.dd + 1;
  const int ext_h = (TH - 1) * p.sh + (p.kh - 1) * p.dh + 1;
  const int ext_w = (TW - 1) * p.sw + (p.kw - 1) * p.dw + 1;
  const int chanStride = ext_d * ext_h * ext_w;
  float* s_x = smem;                       // [CCH][chanStride]
  float* s_g = s_x + CCH * chanStride;     // [K][TH*TW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cchunk = blockIdx.x / nblk, pchunk = blockIdx.x % nblk;
  const int c0 = cchunk * CCH;
  const int ncc = min(CCH, p.C - c0);
  const int npair = ncc * p.T;
  // up to two (cc, tap) pairs per thread (CCH * T <= 512)
  int base[2];
  bool ok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int pr = tid + 256 * u;
    ok[u] = pr < npair;
    const int q = ok[u] ? pr : 0;
    const int cc = q / p.T, t = q - cc * p.T;
    const int tw_ = t % p.kw, th_ = (t / p.kw) % p.kh, td_ = t / (p.kw * p.kh);
    base[u] = cc * chanStride + (td_ * p.dd * ext_h + th_ * p.dh) * ext_w + tw_ * p.dw;
  }
  float acc[2][MAXK];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[u][k] = 0.f;
  const long long xvol = (long long)p.ID * p.IH * p.IW, gvol = (long long)p.OD * p.OH * p.OW;
  const int rows_per_chan = ext_d * ext_h;
  for (long long tile = pchunk; tile < ntiles; tile += nblk) {
    long long b = tile;
    const int tw = (int)(b % tilesW); b /= tilesW;
    const int th = (int)(b % tilesH); b /= tilesH;
    const int qd = (int)(b % p.OD);
    const int n = (int)(b / p.OD);
    const int q0h = th * TH, q0w = tw * TW;
    const int i0d = qd * p.sd - p.pd, i0h = q0h * p.sh - p.ph, i0w = q0w * p.sw - p.pw;
    __syncthreads();
    const float* xn = x + ((long long)n * p.C + c0) * xvol;
    for (int rowid = wave; rowid < ncc * rows_per_chan; rowid += 4) {
      const int cc = rowid / rows_per_chan;
      const int rem = rowid - cc * rows_per_chan;
      const int pl = rem / ext_h, rr = rem - pl * ext_h;
      const int id = i0d + pl, ih = i0h + rr;
      const bool rowok = id >= 0 && id < p.ID && ih >= 0 && ih < p.IH;
      const float* src = xn + (long long)cc * xvol + ((long long)id * p.IH + ih) * p.IW;
      float* dst = s_x + cc * chanStride + rem * ext_w;
      for (int col = lane; col < ext_w; col += 64) {
        const int iw = i0w + col;
        dst[col] = (rowok && iw >= 0 && iw < p.IW) ? src[iw] : 0.f;
      }
    }
    for (int i = tid; i < p.K * TH * TW; i += 256) {
      const int k = i / (TH * TW), r = (i / TW) % TH, cx = i % TW;
      const int qh = q0h + r, qw = q0w + cx;
      s_g[i] = (qh < p.OH && qw < p.OW) ? g[((long long)n * p.K + k) * gvol + ((long long)qd * p.OH + qh) * p.OW + qw] : 0.f;
    }
    __syncthreads();
    for (int r = 0; r < TH; ++r) {
      for (int cx = 0; cx < TW; ++cx) {
        const int posoff = (r * p.sh) * ext_w + cx * p.sw;
        float gv[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) gv[k] = k < p.K ? s_g[k * TH * TW + r * TW + cx] : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float xv = s_x[base[u] + posoff];
#pragma unroll
          for (int k = 0; k < MAXK; ++k) acc[u][k] = fmaf(gv[k], xv, acc[u][k]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!ok[u]) continue;
    const int pr = tid + 256 * u;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < p.K) atomicAdd(&dw[((long long)k * p.C + c0) * p.T + pr], acc[u][k]);
  }
}

int fill(SkP& p, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
         int dh, int dw) {
  if (N <= 0 || C <= 0 || K <= 0 || K > MAXK) return DPF_ERR_UNSUPPORTED;
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.T = kd * kh * kw;
  p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.OD = (ID + 2 * pd - (dd * (kd - 1) + 1)) / sd + 1;
  p.OH = (IH + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.OW = (IW + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (p.OD <= 0 || p.OH <= 0 || p.OW <= 0 || p.T > 27) return DPF_ERR_INVALID_ARG;
  return DPF_OK;
}

}  // namespace

extern "C" {

// same tensor conventions as dpf_conv_forward; K <= 4 and K*C*T*4 bytes <= 64 KiB
int dpf_conv_smallk_forward(const float* x, const float* w, const float* bias, float* out, int N, int C, int ID, int IH, int IW, int K, int kd,
                            int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, void* stream) {
  dpf_clear_error();
  if (!x || !w || !out) return DPF_ERR_INVALID_ARG;
  SkP p{};
  int rc = fill(p, N, C, ID, IH, IW, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw);
  if (rc != DPF_OK) return rc;
  const size_t lds = sizeof(float) * (size_t)K * C * p.T;
  if (lds > 64 * 1024) return DPF_ERR_UNSUPPORTED;
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)smallk_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return DPF_ERR_LAUNCH;
  const long long total = (long long)N * p.OD * p.OH * p.OW;
  hipLaunchKernelGGL(smallk_fwd_kernel, dim3(dpf_ew_grid(total)), dim3(256), lds, (hipStream_t)stream, x, w, bias, out, p);
  return dpf_check_launch();
}

// dw[K][C][T] += sum g[n,k,q] * x[n,c,q*s - p + t*dil]   (g [N,K,OD,OH,OW], x [N,C,ID,IH,IW])
int dpf_conv_smallk_wgrad(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int sd,
                          int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_, void* stream) {
  dpf_clear_error();
  if (!g || !x || !dw) return DPF_ERR_INVALID_ARG;
  SkP p{};
  int rc = fill(p, N, C, ID, IH, IW, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw_);
  if (rc != DPF_OK) return rc;
  constexpr int CCH = 16;
  if (CCH * p.T > 512) return DPF_ERR_UNSUPPORTED;
  const int ext_d = (kd - 1) * dd + 1, ext_h = (TH - 1) * sh + (kh - 1) * dh + 1, ext_w = (TW - 1) * sw + (kw - 1) * dw_ + 1;
  const size_t lds = sizeof(float) * ((size_t)CCH * ext_d * ext_h * ext_w + (size_t)K * TH * TW);
  if (lds > 150 * 1024) return DPF_ERR_UNSUPPORTED;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)smallk_wgrad_kernel<CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return DPF_ERR_LAUNCH;
  const int tilesH = dpf_div_up(p.OH, TH), tilesW = dpf_div_up(p.OW, TW);
  const long long ntiles = (long long)N * p.OD * tilesH * tilesW;
  const int cchunks = dpf_div_up(C, CCH);
  long long nblk = 1024 / cchunks;
  if (nblk < 1) nblk = 1;
  if (nblk > ntiles) nblk = ntiles;
  hipLaunchKernelGGL((smallk_wgrad_kernel<CCH>), dim3((unsigned)(cchunks * nblk)), dim3(256), lds, (hipStream_t)stream, g, x, dw, p, (int)nblk,
                     tilesH, tilesW, ntiles);
  return dpf_check_launch();
}

}  // extern "C"
