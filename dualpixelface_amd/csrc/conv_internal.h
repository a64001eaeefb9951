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
// operand precision "f32" (dpf_set_f32_matrix_path / DPF_F32_X9): 0 = v_mfma_f32_*, 1 = six bf16 partial products of exact three-way splits,
// 2 = three f16 partial products of block-scaled two-way splits
int dpf_conv_f32_x9();
int dpf_h3_range_guard();              // 1 (always, outside tests): the range guards of the f16-component path are active (dpf_debug_set_range_guard)
inline int dpf_conv_f32_nc() { return dpf_conv_f32_x9() == 2 ? 2 : 3; }      // components per operand of the split paths

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

// ---- fp32 products on the f16 matrix pipe (dpf_set_f32_matrix_path(2)): x * 2^s = hi + lo + e with hi = f16(x 2^s), lo = f16(x 2^s - hi),
// both rounded to nearest: |lo| <= 2^-11 |hi|, |e| <= 2^-22 |x 2^s| (rms 2^-24, zero mean) as long as lo is a normal f16.  THREE partial
// products per pair -- lo*hi, hi*lo, hi*hi; lo*lo <= 2^-22 |xy| 2^-2 is dropped -- on v_mfma_f32_32x32x16_f16: half the matrix-pipe work of
// the six bf16 products.  f16 has 5 exponent bits, so the operands are scaled by a power of two (exact) that puts the largest magnitude of
// the staged block into [2^14, 2^15); the accumulators carry the exponent and are rescaled (exactly) when it changes.  A value more than
// 2^17 (DPF_H3_RANGE) below the scale's maximum has a subnormal lo and loses low bits (absolute error 2^-40 of that maximum), so every
// kernel on this path GUARDS the range, each along the axis its output elements do NOT sum over:
//   * igemm3_x9_kernel (output = sum over channels and taps at a position): per POSITION.  A position of a chunk whose values all lie more
//     than 2^17 below the scale is DEFERRED: it contributes exact zeros to the pass (the matrix core loses accumulator bits when it is fed
//     subnormal f16 values next to large operands -- tools/probes/mfma_f16_accum_probe.hip) and its values stay in registers; a chunk with a
//     non-zero deferred position is contracted again with the same weights, the deferred positions at their own scale: every position is
//     contracted exactly once, in the pass whose scale lies within 2^17 of it (up to DPF_H3_MAXPASS extra passes, the last one takes
//     whatever is left).  The weights carry one exponent per OUTPUT ROW (igemm3_pack_x9h_kernel);
//   * wgrad2_kernel (output = sum over positions for a (g channel, x channel) pair): per CHANNEL.  Every g row and every x channel of the
//     workgroup carries its own running exponent (rows / columns of the MFMA tile may be scaled independently);
//   * the deformable conv's gcol products (output = sum over the output channels of a voxel): per VOXEL.
typedef _Float16 dpf_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dpf_pk_f16(float x, float y) {       // round to nearest even (the f16 rounding mode of the kernel)
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// (x, y) already scaled -> packed f16 pairs (hi, lo): 5 vector instructions
__device__ __forceinline__ void dpf_split_pair_h(float x, float y, unsigned& h, unsigned& l) {
  h = dpf_pk_f16(x, y);
  const dpf_f16x2 hh = __builtin_bit_cast(dpf_f16x2, h);
  const dpf_f32x2 v = {x, y};
  const dpf_f32x2 hf = {(float)hh.x, (float)hh.y};
  const dpf_f32x2 r = v - hf;
  l = dpf_pk_f16(r.x, r.y);
}
// the largest biased exponent (bits 30..23 of |v|) over the wave's lanes; every lane passes the bit pattern of a non-negative float
__device__ __forceinline__ int dpf_wave_max_exp(unsigned bits) {
  unsigned v = bits;
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e /* quad_perm [2,3,0,1] */, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /* row_half_mirror */, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140 /* row_mirror */, 0xf, 0xf, false));
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  const unsigned m = max(max(a, b), max(c, d));
  return (int)(m >> 23);
}
// scale that maps magnitudes with biased exponent <= E into [.., 2^15): 2^(141 - E); E in [14, 254]
__device__ __forceinline__ float dpf_h3_scale(int E) { return __builtin_bit_cast(float, (unsigned)(268 - E) << 23); }
constexpr int DPF_H3_EMIN = 14;
constexpr int DPF_H3_RANGE = 17;     // exponents below the scale's maximum with a normal low component
constexpr int DPF_H3_MAXPASS = 3;    // extra passes over the deferred positions of a chunk (igemm3_x9_kernel)
constexpr int DPF_H3_MAXDROP = 66;   // an accumulator exponent never sits more than this below the tile's running maximum (no overflow)
// the smallest value over the wave's lanes (unsigned compare)
__device__ __forceinline__ unsigned dpf_wave_min_u32(unsigned bits) {
  unsigned v = bits;
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}
// exact remainders of the two-component split of (x, y) (already scaled): (x - hi) - lo, in fp32 (both differences are exact)
__device__ __forceinline__ void dpf_split_residual_h(float& x, float& y) {
  unsigned h, l;
  dpf_split_pair_h(x, y, h, l);
  const dpf_f16x2 hh = __builtin_bit_cast(dpf_f16x2, h), ll = __builtin_bit_cast(dpf_f16x2, l);
  x = (x - (float)hh.x) - (float)ll.x;
  y = (y - (float)hh.y) - (float)ll.y;
}
#endif
