// Adaptive Normal Module glue (reference: src/model/stereodpnet/normal_module.py:80-138,154-167,185-190;
// src/utils/geometry.py:21-45): surface sampling (4 nearest of the 8 disparity levels, sorted), the gathered cost
// slices, the scale-normalised XYZ coordinate volume, and the head's sigmoid / mean-over-samples / [-1,1] mapping.
// All HBM-bound elementwise / gather kernels, lanes along W.
#include "dpf_common.h"

namespace {

constexpr int MAXLV = 16;
struct SelP {
  int B, h, w, H, W, L, K;   // K = dsample_num
  float costrange[MAXLV];
};

// idx [B,K,h,w] int32 ascending, sdisp [B,K,h,w]
__global__ void anm_select_kernel(const float* __restrict__ disp_full, int* __restrict__ idx, float* __restrict__ sdisp, SelP p) {
  const long long hw = (long long)p.h * p.w;
  const long long total = (long long)p.B * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % p.w);
    const int y = (int)((i / p.w) % p.h);
    const int b = (int)(i / hw);
    // F.interpolate(scale_factor=0.25, mode='nearest') * 0.25   (normal_module.py:156)
    int sy = (int)floorf((float)y * 4.0f), sx = (int)floorf((float)x * 4.0f);
    sy = sy < p.H - 1 ? sy : p.H - 1;
    sx = sx < p.W - 1 ? sx : p.W - 1;
    const float d = disp_full[((long long)b * p.H + sy) * p.W + sx] * 0.25f;
    float score[MAXLV];
#pragma unroll
    for (int l = 0; l < MAXLV; ++l) score[l] = l < p.L ? 1.0f / (fabsf(p.costrange[l] - d) + 1e-6f) : -1.f;
    unsigned chosen = 0;
    for (int j = 0; j < p.K; ++j) {   // top-K by score (normal_module.py:130-131)
      int best = -1;
      float bs = -2.f;
#pragma unroll
      for (int l = 0; l < MAXLV; ++l) {
        const bool free_ = !((chosen >> l) & 1u);
        if (l < p.L && free_ && score[l] > bs) { bs = score[l]; best = l; }
      }
      chosen |= 1u << best;
    }
    int j = 0;
#pragma unroll
    for (int l = 0; l < MAXLV; ++l) {   // ascending index order == torch.sort(indices) (:134)
      if (l < p.L && ((chosen >> l) & 1u)) {
        idx[((long long)b * p.K + j) * hw + (long long)y * p.w + x] = l;
        sdisp[((long long)b * p.K + j) * hw + (long long)y * p.w + x] = p.costrange[l];
        ++j;
      }
    }
  }
}

// vol[b, c, j, y, x] = cost[b, c, idx[b,j,y,x], y, x]   for c < C;   vol has CV = C+3 channels
// A thread owns 4 consecutive pixels of one (b, channel group): the K selected levels are read once (int4 per level) and reused for
// ACH channels -- no per-element 64-bit division, 16-byte accesses along the pixel axis.
constexpr int ANM_MAXK = 8, ANM_ACH = 8;
__global__ __launch_bounds__(256) void anm_gather_kernel(const float* __restrict__ cost, const int* __restrict__ idx, float* __restrict__ vol,
                                                         int B, int C, int L, int K, int h, int w, int CV) {
  const long long hw = (long long)h * w;
  const long long q4 = (hw + 3) >> 2;
  const int cgroups = (C + ANM_ACH - 1) / ANM_ACH;
  const long long total = (long long)B * cgroups * q4;
  const bool vec = (hw & 3) == 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = (i % q4) * 4;
    const int cg = (int)((i / q4) % cgroups);
    const int b = (int)(i / (q4 * cgroups));
    int lv[ANM_MAXK][4];
#pragma unroll
    for (int j = 0; j < ANM_MAXK; ++j)
      if (j < K)
#pragma unroll
        for (int e = 0; e < 4; ++e) lv[j][e] = pix + e < hw ? idx[((long long)b * K + j) * hw + pix + e] : 0;
    for (int c = cg * ANM_ACH; c < min(C, (cg + 1) * ANM_ACH); ++c) {
      const float* cb = cost + ((long long)b * C + c) * L * hw + pix;
      float* vb = vol + ((long long)b * CV + c) * K * hw + pix;
#pragma unroll
      for (int j = 0; j < ANM_MAXK; ++j) {
        if (j >= K) break;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pix + e < hw ? cb[(long long)lv[j][e] * hw + e] : 0.f;
        if (vec) {
          *reinterpret_cast<float4*>(vb + (long long)j * hw) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          for (int e = 0; e < 4 && pix + e < hw; ++e) vb[(long long)j * hw + e] = v[e];
        }
      }
    }
  }
}

// dcost[b,c,l,y,x] = sum_j [idx==l] dvol[b,c,j,y,x]: same ownership; the L outputs of a pixel are built in registers from the K
// incoming values (the selected levels are distinct and ascending) and stored as float4 rows.
constexpr int ANM_MAXL = 16;
__global__ __launch_bounds__(256) void anm_gather_bwd_kernel(const float* __restrict__ dvol, const int* __restrict__ idx, float* __restrict__ dcost,
                                                             int B, int C, int L, int K, int h, int w, int CV) {
  const long long hw = (long long)h * w;
  const long long q4 = (hw + 3) >> 2;
  const int cgroups = (C + ANM_ACH - 1) / ANM_ACH;
  const long long total = (long long)B * cgroups * q4;
  const bool vec = (hw & 3) == 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = (i % q4) * 4;
    const int cg = (int)((i / q4) % cgroups);
    const int b = (int)(i / (q4 * cgroups));
    int lv[ANM_MAXK][4];
#pragma unroll
    for (int j = 0; j < ANM_MAXK; ++j)
      if (j < K)
#pragma unroll
        for (int e = 0; e < 4; ++e) lv[j][e] = pix + e < hw ? idx[((long long)b * K + j) * hw + pix + e] : -1;
    for (int c = cg * ANM_ACH; c < min(C, (cg + 1) * ANM_ACH); ++c) {
      const float* gb = dvol + ((long long)b * CV + c) * K * hw + pix;
      float* ob = dcost + ((long long)b * C + c) * L * hw + pix;
      float g[ANM_MAXK][4];
#pragma unroll
      for (int j = 0; j < ANM_MAXK; ++j)
        if (j < K)
#pragma unroll
          for (int e = 0; e < 4; ++e) g[j][e] = pix + e < hw ? gb[(long long)j * hw + e] : 0.f;
      for (int l = 0; l < L; ++l) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < ANM_MAXK; ++j)
          if (j < K)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += lv[j][e] == l ? g[j][e] : 0.f;
        if (vec) {
          *reinterpret_cast<float4*>(ob + (long long)l * hw) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          for (int e = 0; e < 4 && pix + e < hw; ++e) ob[(long long)l * hw + e] = v[e];
        }
      }
    }
  }
}

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__global__ void anm_minmax_init_kernel(unsigned* mm, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; }
}

// unnormalised xyz into vol channels C..C+2, and per-sample min / max (normal_module.py:101-114)
__global__ __launch_bounds__(256) void anm_xyz_kernel(const float* __restrict__ Kmat, const float* __restrict__ ab, const float* __restrict__ sdisp,
                                                      float* __restrict__ vol, unsigned* __restrict__ mm, int C, int K, int h, int w, int CV) {
  __shared__ float smn[4], smx[4];
  const int b = blockIdx.y;
  const long long hw = (long long)h * w;
  // K_imgF = K with the first two rows / 4; inverse by adjugate
  float m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = Kmat[b * 9 + i];
#pragma unroll
  for (int i = 0; i < 6; ++i) m[i] = m[i] / 4.0f;
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const float id = 1.0f / det;
  float inv[9];
  inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  const float a = ab[b * 2 + 1], bb = ab[b * 2 + 0];   // abvalue = [b, a] (geometry.py:35-36)
  float lo = 3.4e38f, hi = -3.4e38f;
  const long long total = (long long)K * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % hw;
    const int j = (int)(i / hw);
    const float xf = (float)(pix % w), yf = (float)(pix / w);
    float depth = a / (sdisp[((long long)b * K + j) * hw + pix] - bb);
    if (isnan(depth) || isinf(depth)) depth = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float ray = inv[3 * r] * xf + inv[3 * r + 1] * yf + inv[3 * r + 2];
      const float v = ray * depth;
      vol[(((long long)b * CV + C + r) * K + j) * hw + pix] = v;
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
  lo = dpf_wave_min(lo);
  hi = dpf_wave_max(hi);
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = lo; smx[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    hi = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    atomicMin(&mm[2 * b], f2ord(lo));
    atomicMax(&mm[2 * b + 1], f2ord(hi));
  }
}

__global__ void anm_xyz_norm_kernel(float* __restrict__ vol, const unsigned* __restrict__ mm, int C, int K, int h, int w, int CV) {
  const int b = blockIdx.y;
  const float lo = ord2f(mm[2 * b]), hi = ord2f(mm[2 * b + 1]);
  const float den = hi - lo + 1e-6f;
  const long long n = 3LL * K * h * w;
  float* base = vol + ((long long)b * CV + C) * K * h * w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) base[i] = (base[i] - lo) / den;
}

// out[b,c,Y,X] = 2 * mean_d sigmoid(u[b*Dn+d, c, Y, X]) - 1   (normal_module.py:69-72,187,190)
__global__ void sigmoid_mean_fwd_kernel(const float* __restrict__ u, float* __restrict__ out, int B, int Dn, long long CS) {
  const long long total = (long long)B * CS;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i % CS;
    const int b = (int)(i / CS);
    float s = 0.f;
    for (int d = 0; d < Dn; ++d) s += 1.f / (1.f + expf(-u[((long long)b * Dn + d) * CS + r]));
    out[i] = (s / (float)Dn) * 2.0f - 1.0f;
  }
}
__global__ void sigmoid_mean_bwd_kernel(const float* __restrict__ u, const float* __restrict__ g, float* __restrict__ du, int B, int Dn, long long CS) {
  const long long total = (long long)B * Dn * CS;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i % CS;
    const int b = (int)(i / (CS * Dn));
    const float s = 1.f / (1.f + expf(-u[i]));
    du[i] = g[(long long)b * CS + r] * (2.0f / (float)Dn) * s * (1.f - s);
  }
}

// y[n, :, s] = x[n, :, s] / max(||x[n, :, s]||_2, eps)   (F.normalize(dim=1), src/model/nnet/normal_module_.py:114)
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, long long S, float eps) {
  const long long total = (long long)N * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / S, s = i - n * S;
    const float* p = x + n * C * S + s;
    float q = 0.f;
    for (int c = 0; c < C; ++c) q += p[c * S] * p[c * S];
    const float d = fmaxf(sqrtf(q), eps);
    float* o = y + n * C * S + s;
    for (int c = 0; c < C; ++c) o[c * S] = p[c * S] / d;
  }
}
// dx = (g - y (y . g)) / max(||x||, eps) where the norm exceeds eps, g / eps below it
__global__ void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx, int N, int C, long long S,
                                  float eps) {
  const long long total = (long long)N * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / S, s = i - n * S;
    const float* p = x + n * C * S + s;
    const float* gp = g + n * C * S + s;
    float q = 0.f, dot = 0.f;
    for (int c = 0; c < C; ++c) {
      q += p[c * S] * p[c * S];
      dot += p[c * S] * gp[c * S];
    }
    const float nrm = sqrtf(q);
    float* o = dx + n * C * S + s;
    if (nrm > eps) {
      for (int c = 0; c < C; ++c) o[c * S] = (gp[c * S] - p[c * S] * (dot / q)) / nrm;
    } else {
      for (int c = 0; c < C; ++c) o[c * S] = gp[c * S] / eps;
    }
  }
}

}  // namespace

extern "C" {

// disp_full [B,H,W] -> idx [B,K,h,w] (int32, ascending), sdisp [B,K,h,w];  costrange: L host floats
int dpf_anm_select(const float* disp_full, int* idx, float* sdisp, const float* costrange_host, int B, int H, int W, int h, int w, int L,
                   int K, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!disp_full || !idx || !sdisp || !costrange_host || B <= 0 || L <= 0 || L > MAXLV || K <= 0 || K > L) return DPF_ERR_INVALID_ARG;
  SelP p;
  p.B = B; p.h = h; p.w = w; p.H = H; p.W = W; p.L = L; p.K = K;
  for (int i = 0; i < MAXLV; ++i) p.costrange[i] = i < L ? costrange_host[i] : 0.f;
  hipLaunchKernelGGL(anm_select_kernel, dim3(dpf_ew_grid((long long)B * h * w)), dim3(256), 0, (hipStream_t)stream, disp_full, idx, sdisp, p);
  return dpf_check_launch();
}

// cost [B,C,L,h,w], idx/sdisp [B,K,h,w], Kmat [B,3,3], abvalue [B,2] -> vol [B,C+3,K,h,w];  mm_ws: 2*B uint32
int dpf_anm_volume_forward(const float* cost, const int* idx, const float* sdisp, const float* Kmat, const float* abvalue, float* vol,
                           unsigned* mm_ws, int B, int C, int L, int K, int h, int w, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!cost || !idx || !sdisp || !Kmat || !abvalue || !vol || !mm_ws || B <= 0 || B > 65535) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int CV = C + 3;
  if (K > ANM_MAXK) return DPF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(anm_gather_kernel, dim3(dpf_ew_grid((long long)B * ((C + ANM_ACH - 1) / ANM_ACH) * (((long long)h * w + 3) / 4))), dim3(256), 0, st, cost,
                     idx, vol, B, C, L, K, h, w, CV);
  hipLaunchKernelGGL(anm_minmax_init_kernel, dim3(dpf_div_up(B, 64)), dim3(64), 0, st, mm_ws, B);
  int gx = dpf_div_up((long long)K * h * w, 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(anm_xyz_kernel, dim3(gx, B), dim3(256), 0, st, Kmat, abvalue, sdisp, vol, mm_ws, C, K, h, w, CV);
  hipLaunchKernelGGL(anm_xyz_norm_kernel, dim3(gx, B), dim3(256), 0, st, vol, mm_ws, C, K, h, w, CV);
  return dpf_check_launch();
}

// dvol [B,C+3,K,h,w] -> dcost [B,C,L,h,w] (fully written)
int dpf_anm_volume_backward(const float* dvol, const int* idx, float* dcost, int B, int C, int L, int K, int h, int w, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!dvol || !idx || !dcost || B <= 0) return DPF_ERR_INVALID_ARG;
  if (K > ANM_MAXK || L > ANM_MAXL) return DPF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(anm_gather_bwd_kernel, dim3(dpf_ew_grid((long long)B * ((C + ANM_ACH - 1) / ANM_ACH) * (((long long)h * w + 3) / 4))), dim3(256), 0,
                     (hipStream_t)stream, dvol, idx, dcost, B, C, L, K, h, w, C + 3);
  return dpf_check_launch();
}

// u [B*Dn, CS] -> out [B, CS]
int dpf_sigmoid_mean_forward(const float* u, float* out, int B, int Dn, long long CS, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!u || !out || B <= 0 || Dn <= 0 || CS <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(sigmoid_mean_fwd_kernel, dim3(dpf_ew_grid((long long)B * CS)), dim3(256), 0, (hipStream_t)stream, u, out, B, Dn, CS);
  return dpf_check_launch();
}
int dpf_sigmoid_mean_backward(const float* u, const float* g, float* du, int B, int Dn, long long CS, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!u || !g || !du || B <= 0 || Dn <= 0 || CS <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(sigmoid_mean_bwd_kernel, dim3(dpf_ew_grid((long long)B * Dn * CS)), dim3(256), 0, (hipStream_t)stream, u, g, du, B, Dn, CS);
  return dpf_check_launch();
}

// Normalised camera-space coordinate volume: channels [choff, choff + 3) of vol [B, CV, K, h, w] = (K_imgF^-1 [u, v, 1]) * depth(sdisp),
// min-max normalised per sample (grid_maker_3d: src/model/stereodpnet/normal_module.py:80-114, src/model/nnet/normal_module_.py:50-87).
// sdisp [B, K, h, w] disparities, Kmat [B, 9], abvalue [B, 2] = [b, a]; mm_ws: 2 * B ints of scratch.
int dpf_xyz_volume(const float* sdisp, const float* Kmat, const float* abvalue, float* vol, int* mm_ws, int B, int choff, int CV, int K, int h,
                   int w, void* stream) {
  dpf_clear_error();
  if (!sdisp || !Kmat || !abvalue || !vol || !mm_ws || B <= 0 || K <= 0 || choff < 0 || choff + 3 > CV) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  unsigned* mm = reinterpret_cast<unsigned*>(mm_ws);
  hipLaunchKernelGGL(anm_minmax_init_kernel, dim3(dpf_div_up(B, 64)), dim3(64), 0, st, mm, B);
  int gx = dpf_div_up((long long)K * h * w, 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(anm_xyz_kernel, dim3(gx, B), dim3(256), 0, st, Kmat, abvalue, sdisp, vol, mm, choff, K, h, w, CV);
  hipLaunchKernelGGL(anm_xyz_norm_kernel, dim3(gx, B), dim3(256), 0, st, vol, mm, choff, K, h, w, CV);
  return dpf_check_launch();
}

// F.normalize(x, dim=1) on x [N, C, S] and its gradient
int dpf_l2_normalize_forward(const float* x, float* y, int N, int C, long long S, float eps, void* stream) {
  dpf_clear_error();
  if (!x || !y || N <= 0 || C <= 0 || S <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(dpf_ew_grid((long long)N * S)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, S, eps);
  return dpf_check_launch();
}
int dpf_l2_normalize_backward(const float* x, const float* g, float* dx, int N, int C, long long S, float eps, void* stream) {
  dpf_clear_error();
  if (!x || !g || !dx || N <= 0 || C <= 0 || S <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(dpf_ew_grid((long long)N * S)), dim3(256), 0, (hipStream_t)stream, x, g, dx, N, C, S, eps);
  return dpf_check_launch();
}

}  // extern "C"
