// Depthwise 3x3 convolution (groups = channels), forward / data-gradient / weight-gradient.
// Replaces nn.Conv2d(nin, nin, 3, padding=1, groups=nin) of depthwise_separable_conv
// (reference: src/module/asm/basics.py:39-58; used by DPBlock.conv5, src/model/stereodpnet/modules.py:31).
// 18 FLOP per output element against 8 bytes of compulsory traffic => HBM-bound; one thread per output
// element, lanes along W, the 3x3 taps come from L1/L2.
#include "dpf_common.h"

namespace {

// y[n,c,y,x] = sum_t w[c][t] * x[n,c,y+ty-pad,x+tx-pad]  (flip = 1: taps mirrored -> data gradient)
__global__ void dw_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, long long NC, int C, int H,
                               int W, int k, int pad, int flip) {
  const long long total = NC * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    const long long nc = i / ((long long)W * H);
    const int c = (int)(nc % C);
    const float* xp = x + nc * H * W;
    const float* wp = w + (long long)c * k * k;
    float acc = 0.f;
    for (int ty = 0; ty < k; ++ty) {
      const int sy = yy + ty - pad;
      if (sy < 0 || sy >= H) continue;
      for (int tx = 0; tx < k; ++tx) {
        const int sx = xx + tx - pad;
        if (sx < 0 || sx >= W) continue;
        const float wv = flip ? wp[(k - 1 - ty) * k + (k - 1 - tx)] : wp[ty * k + tx];
        acc += wv * xp[(long long)sy * W + sx];
      }
    }
    y[i] = acc;
  }
}

// dw[c][t] += sum_{n,y,x} g[n,c,y,x] * x[n,c,y+ty-pad,x+tx-pad];  grid = (chunks, N*C)
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw,
                                                       int C, int H, int W, int pad) {
  __shared__ float sm[4];
  const int row = blockIdx.y;
  const int c = row % C;
  const long long S = (long long)H * W;
  const float* gp = g + (long long)row * S;
  const float* xp = x + (long long)row * S;
  const long long s0 = (long long)blockIdx.x * 4096;
  const long long s1 = min(S, s0 + 4096);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (long long s = s0 + threadIdx.x; s < s1; s += 256) {
    const int xx = (int)(s % W), yy = (int)(s / W);
    const float gv = gp[s];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sy = yy + t / 3 - pad, sx = xx + t % 3 - pad;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W) acc[t] += gv * xp[(long long)sy * W + sx];
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
  hipLaunchKernelGGL(dw_conv_kernel, dim3(dpf_ew_grid((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, w, y,
                     (long long)N * C, C, H, W, k, pad, 0);
  return dpf_check_launch();
}

int dpf_depthwise_conv2d_backward_data(const float* g, const float* w, float* dx, int N, int C, int H, int W, int k, int pad, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !w || !dx || N <= 0 || C <= 0 || 2 * pad != k - 1) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(dw_conv_kernel, dim3(dpf_ew_grid((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, g, w, dx,
                     (long long)N * C, C, H, W, k, pad, 1);
  return dpf_check_launch();
}

// dw [C,1,3,3] += ...   (k must be 3)
int dpf_depthwise_conv2d_backward_weight(const float* g, const float* x, float* dw, int N, int C, int H, int W, int k, int pad, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !x || !dw || N <= 0 || C <= 0 || k != 3 || pad != 1 || (long long)N * C > 65535) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(dw_wgrad_kernel, dim3((unsigned)dpf_div_up((long long)H * W, 4096), (unsigned)(N * C)), dim3(256), 0,
                     (hipStream_t)stream, g, x, dw, C, H, W, pad);
  return dpf_check_launch();
}

}  // extern "C"
