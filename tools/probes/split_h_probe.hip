// dpf_split_pair_h / dpf_split_residual_h on sample values (subnormal components): prints hi, lo and the remainder
// hipcc --offload-arch=gfx950 -O3 -I dualpixelface_amd/csrc -o split_h_probe split_h_probe.hip
#include "conv_internal.h"
#include <cstdio>
#include <cmath>
__global__ void k(const float* v, float* out, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  unsigned h, l;
  dpf_split_pair_h(v[i], -v[i], h, l);
  const dpf_f16x2 hh = __builtin_bit_cast(dpf_f16x2, h), ll = __builtin_bit_cast(dpf_f16x2, l);
  float a = v[i], b = -v[i];
  dpf_split_residual_h(a, b);
  out[4 * i] = (float)hh.x; out[4 * i + 1] = (float)ll.x; out[4 * i + 2] = a; out[4 * i + 3] = (float)ll.y;
}
int main() {
  const int n = 8;
  float hv[n] = {20000.3f, 0.8f, ldexpf(1.2345678f, -7), ldexpf(1.2345678f, -14), ldexpf(1.2345678f, -20), ldexpf(1.7345678f, -3), ldexpf(1.2345678f, -26), 3.0f};
  float *dv, *dout, ho[4 * n];
  (void)hipMalloc(&dv, n * 4); (void)hipMalloc(&dout, 16 * n);
  (void)hipMemcpy(dv, hv, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dv, dout, n);
  (void)hipMemcpy(ho, dout, 16 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("v %.9g = 2^%.2f: hi %.9g lo %.9g (lo of -v %.9g) remainder %.9g (2^%.1f of v)  check v-hi-lo-rem = %.3g\n", hv[i], log2(fabs(hv[i])), ho[4 * i], ho[4 * i + 1], ho[4 * i + 3], ho[4 * i + 2],
           ho[4 * i + 2] != 0 ? log2(fabs(ho[4 * i + 2] / hv[i])) : -999., (double)hv[i] - ho[4 * i] - ho[4 * i + 1] - ho[4 * i + 2]);
  return 0;
}
