// Weight repack shared by the implicit-GEMM kernels:  w[A][B][T] -> wt[T][R][KT]  (out index fastest, zero padded)
//   mode 0: reduce = B, out = A      mode 1: reduce = A, out = B      (o0, ocount): optional sub-range of the out index
#pragma once
#include "dpf_common.h"

namespace {
__global__ void repack_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int A, int B, int T, int KT, int mode,
                                      int o0 = 0, int ocount = -1) {
  const int R = mode == 0 ? B : A;
  const int Ofull = mode == 0 ? A : B;
  const int O = ocount < 0 ? Ofull : ocount;   // out channels [o0, o0 + O) of the full tensor
  const long long total = (long long)T * R * KT;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i % KT);
    const int r = (int)((i / KT) % R);
    const int t = (int)(i / ((long long)KT * R));
    float v = 0.f;
    if (o < O) {
      const int a = mode == 0 ? o0 + o : r;
      const int b = mode == 0 ? r : o0 + o;
      v = w[((long long)a * B + b) * T + t];
    }
    wt[i] = v;
  }
}
}  // namespace
