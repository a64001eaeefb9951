// Resampling kernels on NCHW planes (HBM-bound, one thread per output element, lanes along W).
//
//   dpf_upsample_bilinear2d_{forward,backward}  F.interpolate(mode='bilinear', align_corners=True)
//        (reference: src/model/stereodpnet/modules.py:127-128, normal_module.py:22-29,69-72)
//   dpf_upsample_nearest_add_{forward,backward} lat + F.interpolate(top, size=lat.shape[-2:], mode='nearest')
//        (torchvision FeaturePyramidNetwork top-down path; call site modules.py:83-85,119)
// Backward passes are gather-form (each source element sums the destinations whose footprint
// contains it, using the same float32 index arithmetic as the forward) => deterministic, no atomics.
#include "dpf_common.h"

namespace {

// off = 0: align_corners=True source index ratio * dst (ratio = (in-1)/(out-1)); off = 0.5: align_corners=False, ratio = in/out,
// source index ratio * (dst + 0.5) - 0.5 clamped at 0 (ATen area_pixel_compute_source_index)
__device__ __forceinline__ void ac_src(int dst, float ratio, int in, int& i0, int& i1, float& lam, float off = 0.f) {
  float src = ratio * ((float)dst + off) - off;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = src - (float)i0;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long NC, int h, int w, int H, int W,
                                    float off = 0.f) {
  const float ry = off > 0.f ? (float)h / (float)H : (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f);
  const float rx = off > 0.f ? (float)w / (float)W : (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f);
  const long long total = NC * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    const int Y = (int)((i / W) % H);
    const long long nc = i / ((long long)W * H);
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(Y, ry, h, y0, y1, ly, off);
    ac_src(X, rx, w, x0, x1, lx, off);
    const float* p = x + nc * h * w;
    const float hy = 1.f - ly, hx = 1.f - lx;
    y[i] = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
  }
}

// weight of destination index d on source index i along one axis
__device__ __forceinline__ float ac_weight(int d, float ratio, int in, int i, float off = 0.f) {
  int i0, i1;
  float lam;
  ac_src(d, ratio, in, i0, i1, lam, off);
  float wgt = 0.f;
  if (i0 == i) wgt += 1.f - lam;
  if (i1 == i) wgt += lam;
  return wgt;
}

__device__ __forceinline__ void cand_range(int i, float ratio, int out, int& lo, int& hi, float off = 0.f) {
  if (ratio <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf(((float)(i - 1) + off) / ratio - off) - 1;
  hi = (int)ceilf(((float)(i + 1) + off) / ratio - off) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

__global__ void bilinear_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long NC, int h, int w, int H, int W,
                                    float off = 0.f) {
  const float ry = off > 0.f ? (float)h / (float)H : (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f);
  const float rx = off > 0.f ? (float)w / (float)W : (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f);
  const long long total = NC * h * w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xs = (int)(i % w);
    const int ys = (int)((i / w) % h);
    const long long nc = i / ((long long)w * h);
    int ylo, yhi, xlo, xhi;
    cand_range(ys, ry, H, ylo, yhi, off);
    cand_range(xs, rx, W, xlo, xhi, off);
    const float* gp = g + nc * H * W;
    float acc = 0.f;
    for (int Y = ylo; Y <= yhi; ++Y) {
      const float wy = ac_weight(Y, ry, h, ys, off);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int X = xlo; X <= xhi; ++X) {
        const float wx = ac_weight(X, rx, w, xs, off);
        if (wx != 0.f) row += wx * gp[(long long)Y * W + X];
      }
      acc += wy * row;
    }
    dx[i] = acc;
  }
}

__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
  int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__global__ void nearest_add_fwd_kernel(const float* __restrict__ lat, const float* __restrict__ top, float* __restrict__ y,
                                       long long NC, int h, int w, int H, int W) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long long total = NC * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    const int Y = (int)((i / W) % H);
    const long long nc = i / ((long long)W * H);
    y[i] = lat[i] + top[nc * h * w + (long long)nearest_src(Y, sy, h) * w + nearest_src(X, sx, w)];
  }
}

__global__ void nearest_bwd_kernel(const float* __restrict__ g, float* __restrict__ dtop, long long NC, int h, int w, int H, int W) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long long total = NC * h * w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xs = (int)(i % w);
    const int ys = (int)((i / w) % h);
    const long long nc = i / ((long long)w * h);
    int ylo = (int)floorf((float)ys / sy) - 1, yhi = (int)ceilf((float)(ys + 1) / sy) + 1;
    int xlo = (int)floorf((float)xs / sx) - 1, xhi = (int)ceilf((float)(xs + 1) / sx) + 1;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > H - 1 ? H - 1 : yhi; xhi = xhi > W - 1 ? W - 1 : xhi;
    const float* gp = g + nc * H * W;
    float acc = 0.f;
    for (int Y = ylo; Y <= yhi; ++Y) {
      if (nearest_src(Y, sy, h) != ys) continue;
      for (int X = xlo; X <= xhi; ++X)
        if (nearest_src(X, sx, w) == xs) acc += gp[(long long)Y * W + X];
    }
    dtop[i] = acc;
  }
}

// nn.AvgPool2d(k, stride = k): one wave per output element, lanes stride over the k x k window (PSMNet's SPP branches pool
// 64 / 32 / 16 / 8 windows of the quarter-resolution feature map, psmnet/modules.py:84-102)
__global__ __launch_bounds__(256) void avg_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long NC, int H, int W, int k, int OH,
                                                           int OW) {
  const int lane = threadIdx.x & 63;
  const long long total = NC * OH * OW;
  for (long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); o < total; o += (long long)gridDim.x * 4) {
    const int ox = (int)(o % OW), oy = (int)((o / OW) % OH);
    const long long nc = o / ((long long)OW * OH);
    const float* src = x + (nc * H + (long long)oy * k) * W + (long long)ox * k;
    float a = 0.f;
    for (int e = lane; e < k * k; e += 64) a += src[(long long)(e / k) * W + (e % k)];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) a += __shfl_xor(a, s);
    if (lane == 0) y[o] = a / (float)(k * k);
  }
}

__global__ __launch_bounds__(256) void avg_pool_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long NC, int H, int W, int k, int OH,
                                                           int OW) {
  const long long total = NC * H * W;
  const float inv = 1.f / (float)(k * k);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xx = (int)(i % W), yy = (int)((i / W) % H);
    const long long nc = i / ((long long)W * H);
    const int oy = yy / k, ox = xx / k;
    dx[i] = (oy < OH && ox < OW) ? g[(nc * OH + oy) * OW + ox] * inv : 0.f;
  }
}

}  // namespace

extern "C" {

int dpf_upsample_bilinear2d_forward(const float* x, float* y, long long NC, int h, int w, int H, int W, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !y || NC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(dpf_ew_grid(NC * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, NC, h, w, H, W, 0.f);
  return dpf_check_launch();
}

int dpf_upsample_bilinear2d_backward(const float* g, float* dx, long long NC, int h, int w, int H, int W, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !dx || NC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(dpf_ew_grid(NC * h * w)), dim3(256), 0, (hipStream_t)stream, g, dx, NC, h, w, H, W, 0.f);
  return dpf_check_launch();
}

// F.interpolate(x, size=(H, W), mode='bilinear', align_corners=<flag>): the half-pixel convention (align_corners=False) is what the
// NNet feature extractor's pyramid branches use (src/model/nnet/modules.py:110-120)
int dpf_resize_bilinear2d_forward(const float* x, float* y, long long NC, int h, int w, int H, int W, int align_corners, void* stream) {
  dpf_clear_error();
  if (!x || !y || NC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(dpf_ew_grid(NC * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, NC, h, w, H, W,
                     align_corners ? 0.f : 0.5f);
  return dpf_check_launch();
}

int dpf_resize_bilinear2d_backward(const float* g, float* dx, long long NC, int h, int w, int H, int W, int align_corners, void* stream) {
  dpf_clear_error();
  if (!g || !dx || NC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(dpf_ew_grid(NC * h * w)), dim3(256), 0, (hipStream_t)stream, g, dx, NC, h, w, H, W,
                     align_corners ? 0.f : 0.5f);
  return dpf_check_launch();
}

// y[NC,H,W] = lat[NC,H,W] + nearest_upsample(top[NC,h,w])
int dpf_upsample_nearest_add_forward(const float* lat, const float* top, float* y, long long NC, int h, int w, int H, int W, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!lat || !top || !y || NC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(nearest_add_fwd_kernel, dim3(dpf_ew_grid(NC * H * W)), dim3(256), 0, (hipStream_t)stream, lat, top, y, NC, h, w, H, W);
  return dpf_check_launch();
}

// dtop[NC,h,w] = adjoint of the nearest upsample applied to g[NC,H,W]  (d lat = g itself)
int dpf_upsample_nearest_backward(const float* g, float* dtop, long long NC, int h, int w, int H, int W, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !dtop || NC <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(nearest_bwd_kernel, dim3(dpf_ew_grid(NC * h * w)), dim3(256), 0, (hipStream_t)stream, g, dtop, NC, h, w, H, W);
  return dpf_check_launch();
}

// y[NC, H/k, W/k] = AvgPool2d(k, stride k)(x[NC, H, W]) (floor mode, no padding)
int dpf_avg_pool2d_forward(const float* x, float* y, long long NC, int H, int W, int k, void* stream) {
  dpf_clear_error();
  if (!x || !y || NC <= 0 || k <= 0 || H < k || W < k) return DPF_ERR_INVALID_ARG;
  const int OH = H / k, OW = W / k;
  const long long total = NC * OH * OW;
  hipLaunchKernelGGL(avg_pool_fwd_kernel, dim3(dpf_ew_grid(total, 4)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, k, OH, OW);
  return dpf_check_launch();
}

int dpf_avg_pool2d_backward(const float* g, float* dx, long long NC, int H, int W, int k, void* stream) {
  dpf_clear_error();
  if (!g || !dx || NC <= 0 || k <= 0 || H < k || W < k) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(avg_pool_bwd_kernel, dim3(dpf_ew_grid(NC * H * W)), dim3(256), 0, (hipStream_t)stream, g, dx, NC, H, W, k, H / k, W / k);
  return dpf_check_launch();
}

}  // extern "C"
