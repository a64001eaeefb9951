// Internal (non-ABI) interface between the deformable-convolution translation units.
#pragma once
#include "dpf_common.h"

// replicas of the grad_weight scratch tensor dwtmp[rep][27][nchunk][64][16] that the backward kernels of BOTH translation units add into and
// dcn_wgrad_fold_kernel (dcn3d.hip) folds
constexpr int DCN_WG_NREP = 8;

// "Lean" kernels (dcn_lean.hip) for the configuration StereoDPNet uses: 3x3x3 taps, stride 1, padding 1, dilation 1, depth <= 4,
// rows 16-byte aligned (W % 4 == 0), K <= 64.  Each returns DPF_ERR_UNSUPPORTED when the shape is not eligible (the caller then uses the
// generic region kernels of dcn3d.hip), DPF_OK when it launched.
//
// weight: the caller's [K][C][27] tensor; ws: workspace of at least dcn_lean_workspace_floats(C, K) floats (weights repacked into the
// matrix waves' fragment order; the larger of the forward and the backward repack).
int dcn_lean_chunk(int C);                              // channel-chunk width (12 or 16) the lean kernels run with
long long dcn_lean_workspace_floats(int C, int K);
int dcn_lean_forward(const float* x, const float* offset, const float* weight, const float* bias, float* out, float* ws, int B, int C, int D, int H,
                     int W, int K, hipStream_t st);
// grad_offset + grad_weight partials: dwtmp[8][27][nchunk][64][16] (zero-initialised by the caller; chunk width dcn_lean_chunk(C)).
int dcn_lean_bwd_offset(const float* x, const float* offset, const float* weight, const float* go, float* doff, float* dwtmp, float* ws, int B, int C,
                        int D, int H, int W, int K, hipStream_t st);
