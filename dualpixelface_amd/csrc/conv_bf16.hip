// 2-D convolutions with bf16 operands and fp32 accumulation / storage on v_mfma_f32_32x32x16_bf16 (BASELINE config 5:
// "MFMA bf16 2D convs, fp32 cost-volume accumulate").  Used for the feature extractor and the ANM 2-D stack when the plugin runs
// with precision = "bf16" (the reference's counterpart is PL's `precision: 16` autocast, config_/train_faceDP.json); activations
// and weights stay fp32 in HBM and are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they are staged, so the result equals an
// fp32 convolution of the bf16-rounded operands up to summation order.
//
//   D[row = out channel][col = 32 consecutive W positions], reduction k = 16 input channels per MFMA.
//   LDS image of a 16-channel chunk of the haloed input tile: [position][16] bf16 (32 B per position), so the B fragment of a lane
//   (position lane&31, channels 8*(lane>>5) .. +7) is ONE ds_read_b128 and a wave's 64 reads cover 2 KiB contiguously.
//   Weights are repacked per launch to bf16 [tap][chunk][out channel][16]: the A fragment is one 16-B global (L2) load.
// Forward handles stride 1 / 2 and any dilation; the data gradient of stride-1 convs is the same kernel on grad_output with the
// taps flipped and the channel roles swapped (strided data gradients and all weight gradients stay on the fp32 kernels).
#include "dpf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TW = 32;

struct CbP {
  int N, C, K;          // images, reduce channels, output channels of this launch
  int IH, IW, OH, OW;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  int nchunk;           // ceil(C / 16)
  int group;            // chunks staged per barrier pair
  int Ktot, k0;         // the output tensor has Ktot channels; this launch writes [k0, k0 + K)
  int tilesH, tilesW;
};

// w[A][B][kh][kw] fp32 -> wt[tap][chunk][KT][16] bf16.  mode 0 (forward): row = a (out channel), reduce = b.
// mode 1 (data gradient): row = b, reduce = a, taps flipped.  Rows [r0, r0 + KT) of the row index; zero padding elsewhere.
__global__ void repack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int A, int B, int T, int KT, int nchunk, int mode,
                                   int r0) {
  const int R = mode == 0 ? B : A;       // reduce extent
  const int O = mode == 0 ? A : B;       // row extent
  const long long total = (long long)T * nchunk * KT * 16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 15);
    const int row = (int)((i >> 4) % KT);
    const int chunk = (int)((i / (16LL * KT)) % nchunk);
    const int t = (int)(i / (16LL * KT * nchunk));
    const int red = chunk * 16 + j, o = r0 + row;
    float v = 0.f;
    if (red < R && o < O) {
      const int a = mode == 0 ? o : red, b = mode == 0 ? red : o;
      const int ts = mode == 0 ? t : T - 1 - t;
      v = w[((long long)a * B + b) * T + ts];
    }
    wt[i] = (__bf16)v;
  }
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void conv2d_bf16_kernel(const float* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
                                                          float* __restrict__ out, CbP p) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* s_in = reinterpret_cast<__bf16*>(smem_raw);      // [group][ext_h * ext_w][16]
  constexpr int TH = 4 * NT, KT = 32 * MT;
  const int ext_h = (TH - 1) * p.sh + (p.kh - 1) * p.dh + 1;
  const int ext_w = (TW - 1) * p.sw + (p.kw - 1) * p.dw + 1;
  const int npos = ext_h * ext_w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  int b = blockIdx.x;
  const int tw = b % p.tilesW; b /= p.tilesW;
  const int th = b % p.tilesH;
  const int n = b / p.tilesH;
  const int q0h = th * TH, q0w = tw * TW;
  const int i0h = q0h * p.sh - p.ph, i0w = q0w * p.sw - p.pw;
  const long long plane = (long long)p.IH * p.IW;
  const float* xn = x + (long long)n * p.C * plane;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;

  // B-fragment base of each row tile of this wave (elements): position (row, col = l31 * sw), channel half hh
  int bbase[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bbase[t] = (((wave * NT + t) * p.sh) * ext_w + l31 * p.sw) * 16 + 8 * hh;

  // channel chunks are staged G at a time (one barrier pair and one burst of loads per G * 16 channels)
  const int G = p.group;
  for (int chunk0 = 0; chunk0 < p.nchunk; chunk0 += G) {
    const int ng = min(G, p.nchunk - chunk0);
    __syncthreads();                                  // previous group consumed
    // ---- stage: item = (channel octet, position); 8 coalesced row loads (one per channel), one 16-B LDS write
    const int c0 = chunk0 * 16;
    for (int i = tid; i < 2 * ng * npos; i += 256) {
      const int o = i / npos;                          // channel octet within the group
      const int pos = i - o * npos;
      const int r = pos / ext_w, cc = pos - r * ext_w;
      const int ih = i0h + r, iw = i0w + cc;
      const bool inb = ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
      const float* src = xn + (long long)(c0 + 8 * o) * plane + (long long)ih * p.IW + iw;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (inb && c0 + 8 * o + j < p.C) ? src[(long long)j * plane] : 0.f;
      bf16x8 pk;
#pragma unroll
      for (int j = 0; j < 8; ++j) pk[j] = (__bf16)v[j];
      *reinterpret_cast<bf16x8*>(s_in + ((o >> 1) * npos + pos) * 16 + 8 * (o & 1)) = pk;
    }
    __syncthreads();
    // ---- MFMA over the chunks of the group and their taps
    for (int cg = 0; cg < ng; ++cg) {
      const __bf16* wc = wt + ((long long)(chunk0 + cg) * KT + l31) * 16 + 8 * hh;
      const __bf16* sc = s_in + cg * npos * 16;
      for (int a = 0; a < p.kh; ++a) {
        for (int c = 0; c < p.kw; ++c) {
          const int tap = a * p.kw + c;
          const __bf16* wa = wc + (long long)tap * p.nchunk * KT * 16;
          bf16x8 af[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const bf16x8*>(wa + m * 32 * 16);
          const int toff = (a * p.dh * ext_w + c * p.dw) * 16;
          bf16x8 bf[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(sc + bbase[t] + toff);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[t], acc[m][t], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue: D row = (j&3) + 8*(j>>2) + 4*(lane>>5), col = lane&31
  const int ow = q0w + l31;
  const long long oplane = (long long)p.OH * p.OW;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int oh = q0h + wave * NT + t;
    if (oh >= p.OH || ow >= p.OW) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (k < p.K) {
          float v = acc[m][t][j];
          if (bias) v += bias[p.k0 + k];
          out[((long long)n * p.Ktot + p.k0 + k) * oplane + (long long)oh * p.OW + ow] = v;
        }
      }
  }
}

int out_dim(int I, int k, int s, int pd, int d) { return (I + 2 * pd - (d * (k - 1) + 1)) / s + 1; }

// one convolution in "forward form": x has C channels, the output Ktot; weights w[A][B][kh][kw] with (mode 0) A = Ktot, B = C or
// (mode 1) A = C, B = Ktot
int launch(const float* x, const float* w, const float* bias, float* out, void* ws, int N, int C, int IH, int IW, int Ktot, int OH, int OW, int kh,
           int kw, int sh, int sw, int ph, int pw, int dh, int dw, int mode, hipStream_t st) {
  CbP p{};
  p.N = N; p.C = C; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
  p.nchunk = (C + 15) / 16;
  p.Ktot = Ktot;
  const int T = kh * kw;
  const int A = mode == 0 ? Ktot : C, B = mode == 0 ? C : Ktot;
  for (int k0 = 0; k0 < Ktot; k0 += 128) {
    p.k0 = k0;
    p.K = Ktot - k0 < 128 ? Ktot - k0 : 128;
    const int MT = (p.K + 31) / 32, KT = 32 * MT;
    const int NT = MT == 1 ? 4 : 2;
    const int TH = 4 * NT;
    const int ext_h = (TH - 1) * sh + (kh - 1) * dh + 1, ext_w = (TW - 1) * sw + (kw - 1) * dw + 1;
    int group = p.nchunk < 4 ? p.nchunk : 4;
    while (group > 1 && (size_t)group * ext_h * ext_w * 32 > 40 * 1024) --group;     // keep >= 3 workgroups per CU
    p.group = group;
    const size_t lds = (size_t)group * ext_h * ext_w * 32;
    if (lds > 150 * 1024) return DPF_ERR_UNSUPPORTED;
    p.tilesH = dpf_div_up(OH, TH);
    p.tilesW = dpf_div_up(OW, TW);
    const long long blocks = (long long)N * p.tilesH * p.tilesW;
    if (blocks >= 0x7fffffffLL) return DPF_ERR_UNSUPPORTED;
    const long long welems = (long long)T * p.nchunk * KT * 16;
    hipLaunchKernelGGL(repack_bf16_kernel, dim3(dpf_ew_grid(welems)), dim3(256), 0, st, w, (__bf16*)ws, A, B, T, KT, p.nchunk, mode, k0);
#define DPF_CB(M, Nt)                                                                                                           \
  {                                                                                                                             \
    if (lds > 48 * 1024 &&                                                                                                      \
        hipFuncSetAttribute((const void*)conv2d_bf16_kernel<M, Nt>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return DPF_ERR_LAUNCH;                                                                                                    \
    hipLaunchKernelGGL((conv2d_bf16_kernel<M, Nt>), dim3((unsigned)blocks), dim3(256), lds, st, x, (const __bf16*)ws, bias, out, p); \
  }
    switch (MT) { case 1: DPF_CB(1, 4); break; case 2: DPF_CB(2, 2); break; case 3: DPF_CB(3, 2); break; default: DPF_CB(4, 2); break; }
#undef DPF_CB
  }
  return dpf_check_launch();
}

}  // namespace

extern "C" {

// bytes of workspace for dpf_conv2d_bf16_forward / _dgrad (the repacked bf16 weights of one launch)
long long dpf_conv2d_bf16_workspace_bytes(int C, int K, int T) {
  const long long r = C > K ? C : K;
  return (long long)T * ((r + 15) / 16) * 128 * 16 * 2 + 256;
}

// out[N,K,OH,OW] = conv2d(bf16(x[N,C,IH,IW]), bf16(w[K,C,kh,kw])) + bias, fp32 accumulation (nn.Conv2d semantics, groups = 1)
int dpf_conv2d_bf16_forward(const float* x, const float* w, const float* bias, float* out, void* ws, int N, int C, int IH, int IW, int K, int kh,
                            int kw, int sh, int sw, int ph, int pw, int dh, int dw, void* stream) {
  dpf_clear_error();
  if (!x || !w || !out || !ws || N <= 0 || C <= 0 || K <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0)
    return DPF_ERR_INVALID_ARG;
  const int OH = out_dim(IH, kh, sh, ph, dh), OW = out_dim(IW, kw, sw, pw, dw);
  if (OH <= 0 || OW <= 0) return DPF_ERR_INVALID_ARG;
  return launch(x, w, bias, out, ws, N, C, IH, IW, K, OH, OW, kh, kw, sh, sw, ph, pw, dh, dw, 0, (hipStream_t)stream);
}

// dx[N,C,IH,IW] = data gradient of the stride-1 convolution above for grad_output go[N,K,OH,OW] (bf16(go), bf16(w) operands)
int dpf_conv2d_bf16_dgrad(const float* go, const float* w, float* dx, void* ws, int N, int C, int IH, int IW, int K, int kh, int kw, int ph, int pw,
                          int dh, int dw, void* stream) {
  dpf_clear_error();
  if (!go || !w || !dx || !ws || N <= 0 || C <= 0 || K <= 0 || kh <= 0 || kw <= 0 || dh <= 0 || dw <= 0) return DPF_ERR_INVALID_ARG;
  const int OH = out_dim(IH, kh, 1, ph, dh), OW = out_dim(IW, kw, 1, pw, dw);
  if (OH <= 0 || OW <= 0) return DPF_ERR_INVALID_ARG;
  const int qh = dh * (kh - 1) - ph, qw = dw * (kw - 1) - pw;     // padding of the flipped-tap convolution over go
  if (qh < 0 || qw < 0) return DPF_ERR_UNSUPPORTED;
  return launch(go, w, nullptr, dx, ws, N, K, OH, OW, C, IH, IW, kh, kw, 1, 1, qh, qw, dh, dw, 1, (hipStream_t)stream);
}

}  // extern "C"
