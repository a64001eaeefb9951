// FaceDP sample preprocessing on the device: what the reference does per sample on the host in its DataLoader workers
// (dataloader/FaceDP/path_reader.py:150-168 read_depth, :196-232 read_disparity, dataloader/preprocess/preprocess.py:46-88
// basic_transform.apply, augmentation.py:62-84 ToTensor, :165-178 Cropper.applier, :236-262 Lighting, :265-297 Normalizer).
//
// All four kernels are HBM-bound byte movers: one pass over the source window, 16-byte stores.
//   dp_stats_*      max over valid pixels of the depth and of the defocus disparity a / depth + b (fp64), two-phase, no atomics
//   dp_targets      depth window -> depth / mask / disp / idepth crops
//   dp_image        u8 HWC window (+ optional 256-entry photometric LUT per channel) -> normalised f32 CHW; rows are staged
//                   through LDS with aligned dword loads because the window's byte offset (3 * x0) is arbitrary
//   dp_hwc_to_chw   f32 HWC window -> CHW (normal / albedo maps)
#include "dpf_common.h"
#include <math.h>

namespace {

constexpr int ST_BLOCKS = 512;   // partial results of the stats reduction (one per workgroup)

struct DpStat {
  double max_depth, max_disp, bad, valid;
};

template <typename T>
__device__ __forceinline__ bool dp_valid(const T* depth, const unsigned char* mask, long long i) {
  return mask ? (mask[i] != 0) : (depth[i] > (T)0);
}

// NaN-propagating max like np.max
__device__ __forceinline__ double dp_max(double a, double b) { return (a != a || b != b) ? (double)NAN : (a > b ? a : b); }

__device__ __forceinline__ double dp_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = dp_max(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double dp_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void dp_stats_partial_kernel(const T* __restrict__ depth, const unsigned char* __restrict__ mask,
                                                               long long n, double a, double b, double* __restrict__ part) {
  double md = -INFINITY, mq = -INFINITY, cnt = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (!dp_valid(depth, mask, i)) continue;
    const double d = (double)depth[i];
    md = dp_max(md, d);
    mq = dp_max(mq, a / d + b);
    cnt += 1.0;
  }
  __shared__ double sm[3][4];
  md = dp_wave_max(md);
  mq = dp_wave_max(mq);
  cnt = dp_wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) {
    sm[0][threadIdx.x >> 6] = md;
    sm[1][threadIdx.x >> 6] = mq;
    sm[2][threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = dp_max(dp_max(sm[0][0], sm[0][1]), dp_max(sm[0][2], sm[0][3]));
    part[blockIdx.x * 3 + 1] = dp_max(dp_max(sm[1][0], sm[1][1]), dp_max(sm[1][2], sm[1][3]));
    part[blockIdx.x * 3 + 2] = (sm[2][0] + sm[2][1]) + (sm[2][2] + sm[2][3]);
  }
}

// one wave folds the partials in a fixed order -> stats[0..3] = max_depth, max_disp, 0 (bad-pixel count, filled by dp_targets), valid
__global__ __launch_bounds__(64) void dp_stats_final_kernel(double* __restrict__ stats, int nblk) {
  const double* part = stats + 4;
  double md = -INFINITY, mq = -INFINITY, cnt = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    md = dp_max(md, part[i * 3 + 0]);
    mq = dp_max(mq, part[i * 3 + 1]);
    cnt += part[i * 3 + 2];
  }
  md = dp_wave_max(md);
  mq = dp_wave_max(mq);
  cnt = dp_wave_sum(cnt);
  if (threadIdx.x == 0) {
    stats[0] = md;
    stats[1] = mq;
    stats[2] = 0.0;
    stats[3] = cnt;
  }
}

// 4 consecutive output pixels per thread.  disp follows the reference bit for bit: fp64 a / depth + b rounded to fp32 once, the
// fill value 50 * max_disp (fp64) for invalid / NaN / Inf pixels; idepth = max_depth / depth in the depth's own type.
template <typename T>
__global__ __launch_bounds__(256) void dp_targets_kernel(const T* __restrict__ depth, const unsigned char* __restrict__ mask,
                                                         double* __restrict__ stats, double a, double b, int W, int y0, int x0, int ch,
                                                         int cw, float* __restrict__ depth_out, float* __restrict__ mask_out,
                                                         float* __restrict__ disp_out, T* __restrict__ idepth_out, int vec) {
  const int quads = (cw + 3) >> 2;
  const long long total = (long long)ch * quads;
  const double max_depth = stats[0];
  const double fill = stats[1] * 50.0;
  const T max_depth_t = (T)max_depth;
  int bad = 0;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(q / quads), x = (int)(q - (long long)y * quads) * 4;
    const long long src = (long long)(y0 + y) * W + x0 + x;
    const long long dst = (long long)y * cw + x;
    float od[4], om[4], oq[4];
    T oi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      od[j] = om[j] = oq[j] = 0.f;
      oi[j] = (T)0;
      if (x + j < cw) {
        const bool v = dp_valid(depth, mask, src + j);
        const T d = depth[src + j];
        double disp = fill;
        if (v) {
          disp = a / (double)d + b;
          if (disp != disp || isinf(disp)) disp = fill;
          oi[j] = max_depth_t / d;
          od[j] = (float)d;
          om[j] = 1.f;
          const double idc = (double)oi[j];
          if (idc != idc || isinf(idc)) ++bad;
        }
        oq[j] = (float)disp;
        if (disp != disp || isinf(disp)) ++bad;     // only when the fill value itself is not finite (empty mask, Inf disparity)
      }
    }
    if (vec && x + 3 < cw) {
      if (depth_out) *reinterpret_cast<float4*>(depth_out + dst) = make_float4(od[0], od[1], od[2], od[3]);
      if (mask_out) *reinterpret_cast<float4*>(mask_out + dst) = make_float4(om[0], om[1], om[2], om[3]);
      if (disp_out) *reinterpret_cast<float4*>(disp_out + dst) = make_float4(oq[0], oq[1], oq[2], oq[3]);
      if (idepth_out) {
#pragma unroll
        for (int j = 0; j < 4; ++j) idepth_out[dst + j] = oi[j];
      }
    } else {
      for (int j = 0; j < 4 && x + j < cw; ++j) {
        if (depth_out) depth_out[dst + j] = od[j];
        if (mask_out) mask_out[dst + j] = om[j];
        if (disp_out) disp_out[dst + j] = oq[j];
        if (idepth_out) idepth_out[dst + j] = oi[j];
      }
    }
  }
  if (bad) atomicAdd(&stats[2], (double)bad);   // integer-valued, exact in any order
}

struct DpImgPar {
  float shift[4], mean[4], scale[4];
};

// One workgroup = one output row segment of 1024 pixels.  C = 3: 3072 source bytes starting at an arbitrary byte address are
// fetched as aligned dwords into LDS (coalesced), then every thread converts 4 pixels x 3 channels and stores one float4 per plane.
template <int C>
__global__ __launch_bounds__(256) void dp_image_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ lut,
                                                       float* __restrict__ out, long long total_bytes, int W, int y0, int x0, int ch,
                                                       int cw, DpImgPar par, int vec) {
  constexpr int SEG = 1024;
  __shared__ unsigned int stage[(SEG * C + 8) / 4 + 1];
  __shared__ unsigned char slut[C * 256];
  const int y = blockIdx.y;
  const int xs = blockIdx.x * SEG;
  const int npx = min(SEG, cw - xs);
  const long long start = ((long long)(y0 + y) * W + x0 + xs) * C;
  const long long a0 = start & ~3LL;
  const int head = (int)(start - a0);
  const int ndw = (head + npx * C + 3) >> 2;
  for (int i = threadIdx.x; i < ndw; i += 256) {
    const long long addr = a0 + 4LL * i;
    unsigned int v;
    if (addr + 4 <= total_bytes) {
      v = *reinterpret_cast<const unsigned int*>(img + addr);
    } else {                                     // the last dword of the image may be partial
      v = 0;
      for (int k = 0; k < 4 && addr + k < total_bytes; ++k) v |= (unsigned int)img[addr + k] << (8 * k);
    }
    stage[i] = v;
  }
  if (lut)
    for (int i = threadIdx.x; i < C * 256; i += 256) slut[i] = lut[i];
  __syncthreads();
  const unsigned char* sb = reinterpret_cast<const unsigned char*>(stage) + head;
  const int x = threadIdx.x * 4;
  if (x >= npx) return;
  float o[C][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      unsigned int v = (x + j < npx) ? sb[(x + j) * C + c] : 0;
      if (lut) v = slut[c * 256 + v];
      // to_tensor: float(v) / 255 ; Lighting: + shift ; Normalizer: (t - mean) / std -- every step rounded to fp32 as torch does
      float t = (float)v / 255.0f;
      t = t + par.shift[c];
      t = t - par.mean[c];
      o[c][j] = t / par.scale[c];
    }
  }
  const long long plane = (long long)ch * cw;
  const long long dst = (long long)y * cw + xs + x;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if (vec && x + 3 < npx) {
      *reinterpret_cast<float4*>(out + c * plane + dst) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
    } else {
      for (int j = 0; j < 4 && x + j < npx; ++j) out[c * plane + dst + j] = o[c][j];
    }
  }
}

template <int C>
__global__ __launch_bounds__(256) void dp_hwc_to_chw_kernel(const float* __restrict__ src, float* __restrict__ dst, int W, int y0, int x0,
                                                            int ch, int cw, int vec) {
  const int quads = (cw + 3) >> 2;
  const long long total = (long long)ch * quads;
  const long long plane = (long long)ch * cw;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(q / quads), x = (int)(q - (long long)y * quads) * 4;
    const float* s = src + ((long long)(y0 + y) * W + x0 + x) * C;
    float o[C][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < C; ++c) o[c][j] = (x + j < cw) ? s[j * C + c] : 0.f;
    const long long d = (long long)y * cw + x;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (vec && x + 3 < cw) {
        *reinterpret_cast<float4*>(dst + c * plane + d) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
      } else {
        for (int j = 0; j < 4 && x + j < cw; ++j) dst[c * plane + d + j] = o[c][j];
      }
    }
  }
}

bool window_ok(int H, int W, int y0, int x0, int ch, int cw) {
  return H > 0 && W > 0 && ch > 0 && cw > 0 && y0 >= 0 && x0 >= 0 && (long long)y0 + ch <= H && (long long)x0 + cw <= W;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

long long dpf_dp_stats_doubles() { return 4 + 3LL * ST_BLOCKS; }

int dpf_dp_depth_stats(const void* depth, int depth_f64, const unsigned char* mask, long long n, double a, double b, double* stats,
                       void* stream) {
  if (!depth || !stats || n <= 0) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  dpf_clear_error();
  int nblk = dpf_div_up(n, 256 * 8);
  if (nblk > ST_BLOCKS) nblk = ST_BLOCKS;
  if (depth_f64)
    dp_stats_partial_kernel<double><<<nblk, 256, 0, st>>>((const double*)depth, mask, n, a, b, stats + 4);
  else
    dp_stats_partial_kernel<float><<<nblk, 256, 0, st>>>((const float*)depth, mask, n, a, b, stats + 4);
  dp_stats_final_kernel<<<1, 64, 0, st>>>(stats, nblk);
  return dpf_check_launch();
}

int dpf_dp_targets(const void* depth, int depth_f64, const unsigned char* mask, double* stats, double a, double b, int H, int W, int y0,
                   int x0, int ch, int cw, float* depth_out, float* mask_out, float* disp_out, void* idepth_out, void* stream) {
  if (!depth || !stats || !window_ok(H, W, y0, x0, ch, cw)) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  dpf_clear_error();
  const int vec = (cw % 4 == 0) && (!depth_out || aligned16(depth_out)) && (!mask_out || aligned16(mask_out)) &&
                  (!disp_out || aligned16(disp_out));
  const long long total = (long long)ch * ((cw + 3) / 4);
  const int grid = dpf_ew_grid(total);
  if (depth_f64)
    dp_targets_kernel<double><<<grid, 256, 0, st>>>((const double*)depth, mask, stats, a, b, W, y0, x0, ch, cw, depth_out, mask_out,
                                                    disp_out, (double*)idepth_out, vec);
  else
    dp_targets_kernel<float><<<grid, 256, 0, st>>>((const float*)depth, mask, stats, a, b, W, y0, x0, ch, cw, depth_out, mask_out,
                                                   disp_out, (float*)idepth_out, vec);
  return dpf_check_launch();
}

int dpf_dp_image(const unsigned char* img, const unsigned char* lut, float* out, int H, int W, int C, int y0, int x0, int ch, int cw,
                 const float* shift_host, const float* mean_host, const float* std_host, void* stream) {
  if (!img || !out || !window_ok(H, W, y0, x0, ch, cw) || (C != 1 && C != 3) || !mean_host || !std_host) return DPF_ERR_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(img) & 3) return DPF_ERR_INVALID_ARG;   // rows are fetched as aligned dwords
  hipStream_t st = (hipStream_t)stream;
  dpf_clear_error();
  DpImgPar par;
  for (int c = 0; c < 4; ++c) {
    par.shift[c] = (c < C && shift_host) ? shift_host[c] : 0.f;
    par.mean[c] = c < C ? mean_host[c] : 0.f;
    par.scale[c] = c < C ? std_host[c] : 1.f;
  }
  const int vec = (cw % 4 == 0) && aligned16(out);
  dim3 grid(dpf_div_up(cw, 1024), ch);
  const long long total_bytes = (long long)H * W * C;
  if (C == 3)
    dp_image_kernel<3><<<grid, 256, 0, st>>>(img, lut, out, total_bytes, W, y0, x0, ch, cw, par, vec);
  else
    dp_image_kernel<1><<<grid, 256, 0, st>>>(img, lut, out, total_bytes, W, y0, x0, ch, cw, par, vec);
  return dpf_check_launch();
}

int dpf_dp_hwc_to_chw(const float* src, float* dst, int H, int W, int C, int y0, int x0, int ch, int cw, void* stream) {
  if (!src || !dst || !window_ok(H, W, y0, x0, ch, cw) || (C != 1 && C != 3)) return DPF_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  dpf_clear_error();
  const int vec = (cw % 4 == 0) && aligned16(dst);
  const long long total = (long long)ch * ((cw + 3) / 4);
  const int grid = dpf_ew_grid(total);
  if (C == 3)
    dp_hwc_to_chw_kernel<3><<<grid, 256, 0, st>>>(src, dst, W, y0, x0, ch, cw, vec);
  else
    dp_hwc_to_chw_kernel<1><<<grid, 256, 0, st>>>(src, dst, W, y0, x0, ch, cw, vec);
  return dpf_check_launch();
}

}  // extern "C"
