// Depthwise 3x3 convolution (groups = channels), forward / data-gradient / weight-gradient.
// Replaces nn.Conv2d(nin, nin, 3, padding=1, groups=nin) of depthwise_separable_conv
// (reference: src/module/asm/basics.py:39-58; used by DPBlock.conv5, src/model/stereodpnet/modules.py:31).
// 18 FLOP per output element against 8 bytes of compulsory traffic => HBM-bound; one thread per output
// element, lanes along W, the 3x3 taps come from L1/L2.
#include "dpf_common.h"

namespace {

// y[n,c,y,x] = sum_t w[c][t] * x[n,c,y+ty-1,x+tx-1]  (flip = 1: taps mirrored -> data gradient).  k = 3, pad = 1.
// grid.y = (n, c) plane, grid.x strides over row blocks; lanes along x: no integer division per element, the 9 weights of the
// plane are wave-uniform, every output reads 3 rows x 3 shifted (L1-resident) values.
__global__ __launch_bounds__(256) void dw_conv3_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int C, int H,
                                                       int W, int flip) {
  const int nc = blockIdx.y;
  const int c = nc % C;
  const float* xp = x + (long long)nc * H * W;
  float* yp = y + (long long)nc * H * W;
  float wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = w[(long long)c * 9 + (flip ? 8 - t : t)];
  const int lanes_per_row = (W + 255) / 256;             // 256-thread segments per row
  const int nseg = H * lanes_per_row;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int yy = seg / lanes_per_row;
    const int xx = (seg - yy * lanes_per_row) * 256 + threadIdx.x;
    if (xx >= W) continue;
    float acc = 0.f;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int sy = yy + ty - 1;
      if (sy < 0 || sy >= H) continue;
      const float* row = xp + (long long)sy * W;
      const float l = xx > 0 ? row[xx - 1] : 0.f, m = row[xx], r = xx + 1 < W ? row[xx + 1] : 0.f;
      acc = fmaf(wv[ty * 3], l, fmaf(wv[ty * 3 + 1], m, fmaf(wv[ty * 3 + 2], r, acc)));
    }
    yp[(long long)yy * W + xx] = acc;
  }
}

// dw[c][t] += sum_{n,y,x} g[n,c,y,x] * x[n,c,y+ty-1,x+tx-1];  grid = (row blocks, N*C): a block walks whole rows, lanes along x
// nrows > 1 (deterministic mode, dpf_common.h): grid = (1, C) and rows_per_block = H -- one workgroup per channel walks every sample, so
// each dw address receives a single atomic add
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw,
                                                       int C, int H, int W, int rows_per_block, int nrows) {
  __shared__ float sm[4];
  const int c = blockIdx.y % C;
  const long long S = (long long)H * W;
  const int y0 = blockIdx.x * rows_per_block, y1 = min(H, y0 + rows_per_block);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int row = blockIdx.y; row < (nrows > 1 ? nrows : blockIdx.y + 1); row += C)
  for (int yy = y0; yy < y1; ++yy) {
    const float* gp = g + (long long)row * S;
    const float* xp = x + (long long)row * S;
    for (int xx = threadIdx.x; xx < W; xx += 256) {
      const float gv = gp[(long long)yy * W + xx];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int sy = yy + ty - 1;
        if (sy < 0 || sy >= H) continue;
        const float* r = xp + (long long)sy * W;
        acc[ty * 3] += gv * (xx > 0 ? r[xx - 1] : 0.f);
        acc[ty * 3 + 1] += gv * r[xx];
        acc[ty * 3 + 2] += gv * (xx + 1 < W ? r[xx + 1] : 0.f);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float v = dpf_block_sum_256(acc[t], sm);
    if (threadIdx.x == 0) atomicAdd(&dw[c * 9 + t], v);
  }
}

}  // namespace

extern "C" {

// x [N,C,H,W], w [C,1,k,k] -> y [N,C,H,W] (stride 1, dilation 1, padding `pad`)
int dpf_depthwise_conv2d_forward(const float* x, const float* w, float* y, int N, int C, int H, int W, int k, int pad, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !w || !y || N <= 0 || C <= 0 || 2 * pad != k - 1) return DPF_ERR_INVALID_ARG;
  if (k != 3 || (long long)N * C > 65535) return DPF_ERR_UNSUPPORTED;
  {
    const int nseg = H * ((W + 255) / 256);
    int gx = 4096 / (N * C) + 1;
    if (gx > nseg) gx = nseg;
    hipLaunchKernelGGL(dw_conv3_kernel, dim3(gx, N * C), dim3(256), 0, (hipStream_t)stream, x, w, y, C, H, W, 0);
  }
  return dpf_check_launch();
}

int dpf_depthwise_conv2d_backward_data(const float* g, const float* w, float* dx, int N, int C, int H, int W, int k, int pad, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !w || !dx || N <= 0 || C <= 0 || 2 * pad != k - 1) return DPF_ERR_INVALID_ARG;
  if (k != 3 || (long long)N * C > 65535) return DPF_ERR_UNSUPPORTED;
  {
    const int nseg = H * ((W + 255) / 256);
    int gx = 4096 / (N * C) + 1;
    if (gx > nseg) gx = nseg;
    hipLaunchKernelGGL(dw_conv3_kernel, dim3(gx, N * C), dim3(256), 0, (hipStream_t)stream, g, w, dx, C, H, W, 1);
  }
  return dpf_check_launch();
}

// dw [C,1,3,3] += ...   (k must be 3)
int dpf_depthwise_conv2d_backward_weight(const float* g, const float* x, float* dw, int N, int C, int H, int W, int k, int pad, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !x || !dw || N <= 0 || C <= 0 || k != 3 || pad != 1 || (long long)N * C > 65535) return DPF_ERR_INVALID_ARG;
  {
    int rpb = dpf_div_up(4096, W);                 // ~4096 elements per block
    if (rpb < 1) rpb = 1;
    if (dpf_deterministic())
      hipLaunchKernelGGL(dw_wgrad_kernel, dim3(1, (unsigned)C), dim3(256), 0, (hipStream_t)stream, g, x, dw, C, H, W, H, N * C);
    else
      hipLaunchKernelGGL(dw_wgrad_kernel, dim3((unsigned)dpf_div_up(H, rpb), (unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, g, x, dw, C, H, W,
                         rpb, 1);
  }
  return dpf_check_launch();
}

}  // extern "C"
