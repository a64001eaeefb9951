// Internal (non-ABI) interface between the dense-convolution translation units.
#pragma once
#include "dpf_common.h"

// One dense convolution launch (forward, or transposed = data gradient / ConvTranspose3d) covering the output channels
// [k0, k0 + K) of a tensor with Ktot channels.  `w` is the caller's weight tensor w[wA][wB][T]; mode 0: reduce = wB, out = wA
// (forward conv layout), mode 1: reduce = wA, out = wB (transposed layout).
struct DpfConvDesc {
  int N, C, K, Ktot, k0;
  int ID, IH, IW, OD, OH, OW;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int transposed;
  int wA, wB, mode;
  int accumulate;               // 1: out += result (data gradients of several consumers of one tensor summed in the epilogue)
};

// operand precision of the dense convolution kernels (dpf_set_conv_operand_precision): 0 = exact fp32, 1 = operands rounded to bf16
// (RNE) in the staging path, fp32 accumulation and storage
int dpf_conv_operand_bf16();
// operand precision "f32": 1 = fp32 products as nine exact bf16 partial products on the bf16 matrix pipe (default), 0 = v_mfma_f32_* (DPF_F32_X9=0)
int dpf_conv_f32_x9();

// LDS-DMA double-buffered implicit GEMM (conv_igemm2.hip).  Returns DPF_OK when it launched, DPF_ERR_UNSUPPORTED when the
// shape is not eligible (the caller then uses the generic kernel), another error code on failure.
// optional per-tile BatchNorm statistics of a forward launch: slab [parts][K][2] doubles (sum, sum of squares of the outputs)
struct DpfConvStats {
  double* slab;
  long long capacity_doubles;
  int parts;                    // out: rows written
};
int dpf_igemm2_conv(const float* x, const float* w, const float* bias, float* out, float* ws, const DpfConvDesc& d, hipStream_t st,
                    DpfConvStats* stats = nullptr);
// floats of workspace dpf_igemm2_conv may use for (T taps, `reduce` reduction channels, `outc` output channels)
long long dpf_igemm2_workspace_floats(int T, int reduce, int outc);

// One weight-gradient launch for the g-channels [k0, k0 + K) of a g tensor with Ktot channels; dw points at row k0 of dW[Ktot][C][T].
struct DpfWgradDesc {
  int N, C, K, Ktot, k0;
  int ID, IH, IW, QD, QH, QW;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
};
// LDS-DMA double-buffered, slab-reduced (deterministic) weight gradient (conv_wgrad2.hip); DPF_ERR_UNSUPPORTED -> caller falls back.
// accumulate = 0: dw is overwritten (no zero-initialisation needed), 1: dw += ...
int dpf_wgrad2(const float* g, const float* x, float* dw, float* ws, long long ws_floats, const DpfWgradDesc& d, int accumulate, hipStream_t st);
long long dpf_wgrad2_workspace_floats(int T, int C, int K);

// Pointwise (1x1x1) convolutions, HBM-bound direct kernels (conv_pointwise.hip); DPF_ERR_UNSUPPORTED -> caller falls back.
int dpf_pointwise_conv(const float* x, const float* w, const float* bias, float* out, const DpfConvDesc& d, hipStream_t st);
int dpf_pointwise_wgrad(const float* g, const float* x, float* dw, float* ws, long long ws_floats, const DpfWgradDesc& d, int accumulate,
                        hipStream_t st);
long long dpf_pointwise_wgrad_workspace_floats(int C, int K);
