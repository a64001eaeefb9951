// Internal (non-ABI) interface between the deformable-convolution translation units.
#pragma once
#include "dpf_common.h"

// replicas of the grad_weight scratch tensor dwtmp[rep][27][nchunk][64][16] that the backward kernels of BOTH translation units add into and
// dcn_wgrad_fold_kernel (dcn3d.hip) folds
constexpr int DCN_WG_NREP = 8;

#ifdef __HIPCC__
// grad_input / grad_weight-scratch accumulation: float atomics, or -- deterministic mode (dpf_common.h) -- order-independent integer pairs in a
// shadow array indexed like the float tensor (`shadow` != nullptr).  The shadow of grad_input lives behind the ordinary workspace
// (dpf_deform_conv3d_backward_workspace_floats), the grad_weight scratch is its own shadow (a replica of int64 pairs fits in 4 of its 8
// float replicas).
__device__ __forceinline__ void dcn_acc_add(float* base, long long* shadow, float* addr, float v) {
  if (shadow) dpf_det_add(shadow + 2 * (addr - base), v);
  else atomicAdd(addr, v);
}
#endif

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
// det != 0: the partials are added as integer pairs into replica 0 read as long long [27][nchunk][64][16][2] (deterministic mode).
int dcn_lean_bwd_offset(const float* x, const float* offset, const float* weight, const float* go, float* doff, float* dwtmp, float* ws, int B, int C,
                        int D, int H, int W, int K, hipStream_t st, int det = 0);
