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

// ---- fp32 products on the bf16 matrix pipe: the operand split shared by igemm3_x9_kernel, its weight pack kernel and wgrad2_kernel<.., X9>.
// x = hi + mid + lo EXACTLY, by rounding to nearest: hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (the residuals are exact in fp32 and
// the last one has at most 8 significant bits, so its conversion is exact too).  |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, residuals signed.
// Of the nine partial products of a pair the SIX that can reach 2^-24 of the product are issued -- hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
// lo*hi; mid*lo + lo*mid <= 2^-23 |xy| worst case (rms 2^-26, zero mean: the split rounds to nearest) and lo*lo <= 2^-32 |xy| are dropped:
// below the rounding of the fp32 accumulation every product goes into (sum of 864 positive products: dropped terms 1e-9 of the sum, the
// fp32 accumulation itself 7e-7).  DPF_X9_FIRST = 1 issues eight (drops lo*lo only), 0 all nine.
#ifdef __HIPCC__
#ifndef DPF_X9_FIRST
#define DPF_X9_FIRST 3
#endif
typedef __bf16 dpf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dpf_f32x2 __attribute__((ext_vector_type(2)));
// two floats -> two bf16 (round to nearest even) in one register, x in the low half: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned dpf_pk_bf16(float x, float y) {
  const dpf_f32x2 f = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, dpf_bf16x2));
}
// (x, y) -> packed bf16 pairs of the three components: 9 vector instructions (3 v_cvt_pk_bf16_f32, 4 bit operations that widen a packed pair
// back to two floats, 2 v_pk_add_f32 -- the two residuals of a stage are ONE packed subtraction)
__device__ __forceinline__ void dpf_split_pair(float x, float y, unsigned& h, unsigned& m, unsigned& l) {
  const dpf_f32x2 v = {x, y};
  h = dpf_pk_bf16(x, y);
  const dpf_f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
  const dpf_f32x2 r1 = v - hf;
  m = dpf_pk_bf16(r1.x, r1.y);
  const dpf_f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
  const dpf_f32x2 r2 = r1 - mf;
  l = dpf_pk_bf16(r2.x, r2.y);
}
#endif
