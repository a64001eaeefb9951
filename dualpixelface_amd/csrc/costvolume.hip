// Dual-pixel cost-volume construction.
//
//   dpf_shift_triple_{forward,backward}: the three row-shifted copies of a feature map (nearest / bilinear /
//     Fourier-phase) that subpixel_shift.forward produces (reference: src/module/asm/asm.py:87-127), expressed
//     as a table-driven separable 2x2-tap sampler: out[b,c,m,y,x] = sum_{a,e} wy[m][a][y] * wx[m][e][x] *
//     fea[b,c, iy[m][a][y], ix[m][e][x]].  The host builds the tables with the reference's own float32 op
//     order (grid normalisation quirks Q2, un-keyed grid cache Q1); an integer phase shift is a row roll.
//   dpf_cv_select_{forward,backward}: MaskingAttention's tail (asm.py:162-171): softmax over the three copies of
//     sigmoid(mask), weighted mean, written straight into every cost-volume level that shares this shift
//     (CostVolume.build_concat_volume, src/model/stereodpnet/modules.py:181-197).
//   dpf_psm_volume_forward: PSMNet's integer-shift concat / group-wise-correlation volume
//     (src/model/psmnet/modules.py:215-262, int() truncation Q14).
// All HBM-bound; one thread per output element, lanes along W (coalesced 256-B rows per wave).
#include "dpf_common.h"

namespace {

// tables: iy [3][2][h] int32 (-1 = no tap), wy [3][2][h] float, ix [3][2][w], wx [3][2][w]
// grid (row chunks, BC * 3 planes): the plane index (b, c, mode) comes from blockIdx.y, a thread owns 4 consecutive columns of a row:
// one 32-bit division per 4 outputs, the row taps are uniform per row, the column taps are read as vectors, 16-byte stores.
constexpr int ST_ROWS = 8;
__global__ __launch_bounds__(256) void shift_triple_fwd_kernel(const float* __restrict__ fea, float* __restrict__ out, const int* __restrict__ iy,
                                                               const float* __restrict__ wy, const int* __restrict__ ix,
                                                               const float* __restrict__ wx, int h, int w) {
  const int m = blockIdx.y % 3;
  const long long bc = blockIdx.y / 3;
  const int y0 = blockIdx.x * ST_ROWS;
  const int nrow = min(ST_ROWS, h - y0);
  const int w4 = (w + 3) >> 2;
  const float* src = fea + bc * h * w;
  float* dst = out + ((bc * 3 + m) * h + y0) * (long long)w;
  const int* ixm = ix + m * 2 * w;
  const float* wxm = wx + m * 2 * w;
  const bool vec = (w & 3) == 0;
  for (int idx = threadIdx.x; idx < nrow * w4; idx += 256) {
    const int r = idx / w4, x = (idx - r * w4) * 4;
    const int y = y0 + r;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // column taps of these 4 outputs: one 16-byte load per table row when the row length allows it
    int cx[2][4];
    float cw[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (vec) {
        const int4 iv = *reinterpret_cast<const int4*>(ixm + e * w + x);
        const float4 wv = *reinterpret_cast<const float4*>(wxm + e * w + x);
        cx[e][0] = iv.x; cx[e][1] = iv.y; cx[e][2] = iv.z; cx[e][3] = iv.w;
        cw[e][0] = wv.x; cw[e][1] = wv.y; cw[e][2] = wv.z; cw[e][3] = wv.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cx[e][j] = x + j < w ? ixm[e * w + x + j] : -1;
          cw[e][j] = x + j < w ? wxm[e * w + x + j] : 0.f;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = iy[(m * 2 + a) * h + y];
      if (yy < 0) continue;
      const float wa = wy[(m * 2 + a) * h + y];
      if (wa == 0.f) continue;
      const float* row = src + (long long)yy * w;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (cx[e][j] >= 0 && cw[e][j] != 0.f) acc[j] += (wa * cw[e][j]) * row[cx[e][j]];     // zero-weight taps (the second
      }                                                                                         // bilinear column tap) are not fetched
    }
    float* o = dst + (long long)r * w + x;
    if (vec) {
      *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      for (int j = 0; j < 4 && x + j < w; ++j) o[j] = acc[j];
    }
  }
}

// adjoint: dfea (pre-zeroed) += scatter of g[B,C,3,h,w]
__global__ void shift_triple_bwd_kernel(const float* __restrict__ g, float* __restrict__ dfea, const int* __restrict__ iy,
                                        const float* __restrict__ wy, const int* __restrict__ ix, const float* __restrict__ wx,
                                        long long BC, int h, int w) {
  const long long total = BC * 3 * h * w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int m = (int)((i / ((long long)w * h)) % 3);
    const long long bc = i / ((long long)w * h * 3);
    float* dst = dfea + bc * h * w;
    const float gv = g[i];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = iy[(m * 2 + a) * h + y];
      if (yy < 0) continue;
      const float wa = wy[(m * 2 + a) * h + y];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int xx = ix[(m * 2 + e) * w + x];
        if (xx < 0) continue;
        const float wgt = wa * wx[(m * 2 + e) * w + x];
        if (wgt != 0.f) atomicAdd(&dst[(long long)yy * w + xx], wgt * gv);
      }
    }
  }
}

// deterministic adjoint (gather form): iyi / ixi are the INVERSE tables [3][2][2][n] (iyi[m][a][s][yy] = the s-th output row y whose
// tap (m, a) reads source row yy, -1 if none; the tap maps are monotone and at most two outputs share a source -- the host checks
// it), wy / wx stay indexed by the output coordinate.  One thread per dfea element, <= 48 terms in a fixed order, no atomics, no
// pre-zeroing.
__global__ __launch_bounds__(256) void shift_triple_bwd_gather_kernel(const float* __restrict__ g, float* __restrict__ dfea,
                                                                      const int* __restrict__ iyi, const float* __restrict__ wy,
                                                                      const int* __restrict__ ixi, const float* __restrict__ wx, int h, int w) {
  const long long bc = blockIdx.y;
  const int y0 = blockIdx.x * ST_ROWS;
  const int nrow = min(ST_ROWS, h - y0);
  const int w4 = (w + 3) >> 2;
  const long long hw = (long long)h * w;
  const float* gp = g + bc * 3 * hw;
  float* dst = dfea + bc * hw + (long long)y0 * w;
  const bool vec = (w & 3) == 0;
  for (int idx = threadIdx.x; idx < nrow * w4; idx += 256) {
    const int r = idx / w4, xx0 = (idx - r * w4) * 4;
    const int yy = y0 + r;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      // inverse column taps of these 4 source columns and their weights (zero-weight taps are dropped here, once per mode)
      int cx[2][2][4];
      float cw[2][2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const int* tab = ixi + ((m * 2 + e) * 2 + sx) * w + xx0;
          int t4[4];
          if (vec) {
            const int4 iv = *reinterpret_cast<const int4*>(tab);
            t4[0] = iv.x; t4[1] = iv.y; t4[2] = iv.z; t4[3] = iv.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) t4[j] = xx0 + j < w ? tab[j] : -1;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float wv = t4[j] >= 0 ? wx[(m * 2 + e) * w + t4[j]] : 0.f;
            cx[e][sx][j] = wv != 0.f ? t4[j] : -1;
            cw[e][sx][j] = wv;
          }
        }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int sy = 0; sy < 2; ++sy) {
          const int y = iyi[((m * 2 + a) * 2 + sy) * h + yy];
          if (y < 0) continue;
          const float wa = wy[(m * 2 + a) * h + y];
          if (wa == 0.f) continue;
          const float* grow = gp + m * hw + (long long)y * w;
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (cx[e][sx][j] >= 0) acc[j] += (wa * cw[e][sx][j]) * grow[cx[e][sx][j]];
        }
      }
    }
    float* o = dst + (long long)r * w + xx0;
    if (vec) {
      *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      for (int j = 0; j < 4 && xx0 + j < w; ++j) o[j] = acc[j];
    }
  }
}

// ---- fractional Fourier-phase row shift (asm.py:59-75,112-125 with the irfft(onesided=False) semantics) ----------------
// For a shift delta the reference multiplies the 2-D spectrum by M(ky) = exp(2 pi i delta nr(ky) / h) and inverts with a
// complex-to-real transform that only reads the half spectrum kx <= w/2.  M is Hermitian in ky except at the Nyquist row
// (M(h/2) = exp(-i pi delta)), so the result is NOT a pure row interpolation:
//     out[y][x] = sum_y' mr[(y - y') mod h] src[y'][x]  +  scale (-1)^y sum_x' hm[(x - x') mod w] S[x'],   S[x'] = sum_y' (-1)^y' src[y'][x']
// mr = real inverse DFT of M with its Nyquist bin replaced by cos(pi delta), scale = sin(pi delta) / h, hm = the discrete Hilbert
// kernel (2/w) sum_{k=1}^{(w-1)/2} sin(2 pi k d / w) -- the part of the Nyquist row the C2R transform cannot represent leaks into
// a rank-one term.  Tables come from the host (fp64 -> fp32, sampler_tables.py).  The adjoint is the same operator with mr and hm
// index-reversed.
//
// phase_colsum: one workgroup per plane: S, then T[x] = scale * (hm (*) S)[x] -> tbuf[plane][w]
__global__ __launch_bounds__(256) void phase_colsum_kernel(const float* __restrict__ src, long long sps, const float* __restrict__ hm,
                                                           float scale, float* __restrict__ tbuf, int h, int w) {
  extern __shared__ float sm[];
  float* S = sm;            // [w]
  float* hd = sm + w;       // [2w]: hm doubled, so (x - x' + w) needs no modulo
  const float* p = src + (long long)blockIdx.x * sps;
  for (int x = threadIdx.x; x < w; x += 256) {
    float acc = 0.f;
    for (int y = 0; y + 1 < h; y += 2) acc += p[(long long)y * w + x] - p[(long long)(y + 1) * w + x];
    if (h & 1) acc += p[(long long)(h - 1) * w + x];
    S[x] = acc;
    hd[x] = hm[x];
    hd[x + w] = hm[x];
  }
  __syncthreads();
  for (int x = threadIdx.x; x < w; x += 256) {
    float acc = 0.f;
    for (int xp = 0; xp < w; ++xp) acc += hd[x - xp + w] * S[xp];
    tbuf[(long long)blockIdx.x * w + x] = scale * acc;
  }
}

// phase_circ: workgroup = (32-column strip, plane).  The strip [h][32] and the doubled row kernel live in LDS; each wave owns
// 32-row output blocks and contracts D[y][x] = sum_y' mr[(y - y') mod h] strip[y'][x] with v_mfma_f32_32x32x2_f32
// (A: lane&31 = output row, k = source row pair; B: lane&31 = column) -- exact fp32, h/2 MFMAs per block.
__global__ __launch_bounds__(256) void phase_circ_kernel(const float* __restrict__ src, long long sps, float* __restrict__ dst, long long dps,
                                                         const float* __restrict__ mr, const float* __restrict__ tbuf, int h, int w) {
  extern __shared__ float sm[];
  const int hp = (h + 1) & ~1;            // source rows padded to a pair
  float* strip = sm;                      // [hp][32]
  float* md = sm + hp * 32;               // [2h + 64]: md[m] = mr[m mod h]
  const int x0 = blockIdx.x * 32;
  const long long plane = blockIdx.y;
  const float* p = src + plane * sps;
  for (int i = threadIdx.x; i < hp * 32; i += 256) {
    const int y = i >> 5, x = x0 + (i & 31);
    strip[i] = (y < h && x < w) ? p[(long long)y * w + x] : 0.f;
  }
  for (int i = threadIdx.x; i < 2 * h + 64; i += 256) md[i] = mr[i % h];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int x = x0 + l31;
  const float tx = (x < w) ? tbuf[plane * w + x] : 0.f;
  float* d = dst + plane * dps;
  for (int y0 = wv * 32; y0 < h; y0 += 128) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int abase = y0 + l31 + h - hi;          // md index of (y - y') for y' = k + hi at k = 0
    for (int k = 0; k < hp; k += 2) {
      const float a = md[abase - k];
      const float b = strip[(k + hi) * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int y = y0 + (r >> 2) * 8 + hi * 4 + (r & 3);
      if (y < h && x < w) d[(long long)y * w + x] = acc[r] + ((y & 1) ? -tx : tx);
    }
  }
}

// x3, s: [B,C,3,h,w]; vol: [B, CV, L, h, w]; writes channels [choff, choff+C) of every level in `levels` (bit mask)
__global__ void cv_select_fwd_kernel(const float* __restrict__ x3, const float* __restrict__ s, float* __restrict__ vol, int B, int C,
                                     int h, int w, int CV, int L, int choff, unsigned levels) {
  const long long hw = (long long)h * w;
  const long long total = (long long)B * C * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i % hw;
    const long long bc = i / hw;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const long long base = bc * 3 * hw + pix;
    const float s0 = s[base], s1 = s[base + hw], s2 = s[base + 2 * hw];
    const float mx = fmaxf(s0, fmaxf(s1, s2));
    const float e0 = expf(s0 - mx), e1 = expf(s1 - mx), e2 = expf(s2 - mx);
    const float sum = e0 + e1 + e2;
    const float v = (x3[base] * (e0 / sum) + x3[base + hw] * (e1 / sum) + x3[base + 2 * hw] * (e2 / sum)) / 3.0f;
    float* dst = vol + (((long long)b * CV + choff + c) * L) * hw + pix;
    for (int l = 0; l < L; ++l)
      if ((levels >> l) & 1u) dst[(long long)l * hw] = v;
  }
}

__global__ void cv_select_bwd_kernel(const float* __restrict__ x3, const float* __restrict__ s, const float* __restrict__ dvol,
                                     float* __restrict__ dx3, float* __restrict__ ds, int B, int C, int h, int w, int CV, int L,
                                     int choff, unsigned levels) {
  const long long hw = (long long)h * w;
  const long long total = (long long)B * C * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i % hw;
    const long long bc = i / hw;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const long long base = bc * 3 * hw + pix;
    const float* src = dvol + (((long long)b * CV + choff + c) * L) * hw + pix;
    float g = 0.f;
    for (int l = 0; l < L; ++l)
      if ((levels >> l) & 1u) g += src[(long long)l * hw];
    g = g / 3.0f;
    const float s0 = s[base], s1 = s[base + hw], s2 = s[base + 2 * hw];
    const float mx = fmaxf(s0, fmaxf(s1, s2));
    const float e0 = expf(s0 - mx), e1 = expf(s1 - mx), e2 = expf(s2 - mx);
    const float sum = e0 + e1 + e2;
    const float p0 = e0 / sum, p1 = e1 / sum, p2 = e2 / sum;
    const float a0 = x3[base], a1 = x3[base + hw], a2 = x3[base + 2 * hw];
    dx3[base] = p0 * g;
    dx3[base + hw] = p1 * g;
    dx3[base + 2 * hw] = p2 * g;
    const float dp0 = a0 * g, dp1 = a1 * g, dp2 = a2 * g;
    const float dot = p0 * dp0 + p1 * dp1 + p2 * dp2;
    ds[base] = p0 * (dp0 - dot);
    ds[base + hw] = p1 * (dp1 - dot);
    ds[base + 2 * hw] = p2 * (dp2 - dot);
  }
}

struct PsmP {
  int B, C, h, w, L, G;   // G = number of correlation groups (0 = concat only)
  int cv0;                // psm_volume_kernel: first volume channel it computes (2C when the concat channels went through psm_volume_concat_kernel)
  int shift[16];
};

// vol [B, 2C + G, L, h, w]; one workgroup = PSM_RB consecutive rows of one (b, channel, level) plane (contiguous in vol and,
// for the concat channels, in the source feature map too), lanes along x with 16-byte accesses
constexpr int PSM_RB = 8;
__global__ __launch_bounds__(256) void psm_volume_kernel(const float* __restrict__ ref, const float* __restrict__ tar, float* __restrict__ vol, PsmP p) {
  const int CV = 2 * p.C + p.G;
  const int hb = (p.h + PSM_RB - 1) / PSM_RB;
  long long blk = blockIdx.x;
  const int y0 = (int)(blk % hb) * PSM_RB; blk /= hb;
  const int l = (int)(blk % p.L); blk /= p.L;
  const int cv = p.cv0 + (int)(blk % (CV - p.cv0));
  const int b = (int)(blk / (CV - p.cv0));
  const int d = p.shift[l];
  const int nrow = min(PSM_RB, p.h - y0);
  float* dst = vol + ((((long long)b * CV + cv) * p.L + l) * p.h + y0) * p.w;
  const int n = nrow * p.w;
  if (cv < 2 * p.C) {
    const float* src = cv < p.C ? ref + (((long long)b * p.C + cv) * p.h + y0) * p.w
                                : tar + (((long long)b * p.C + (cv - p.C)) * p.h + (y0 + d)) * p.w;
    if ((p.w & 3) == 0) {
      for (int e = 4 * threadIdx.x; e < n; e += 1024) {
        const int y = y0 + e / p.w;
        const bool rowok = d >= 0 ? (y < p.h - d) : (y >= -d);   // rows the reference writes (psmnet/modules.py:229-246)
        *reinterpret_cast<float4*>(dst + e) = rowok ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      for (int e = threadIdx.x; e < n; e += 256) {
        const int y = y0 + e / p.w;
        const bool rowok = d >= 0 ? (y < p.h - d) : (y >= -d);
        dst[e] = rowok ? src[e] : 0.f;
      }
    }
  } else {
    const int gi = cv - 2 * p.C;
    const int cpg = p.C / p.G;
    for (int e = threadIdx.x; e < n; e += 256) {
      const int y = y0 + e / p.w, x = e % p.w;
      const bool rowok = d >= 0 ? (y < p.h - d) : (y >= -d);
      float acc = 0.f;
      if (rowok)
        for (int j = 0; j < cpg; ++j) {
          const int c = gi * cpg + j;
          acc += ref[(((long long)b * p.C + c) * p.h + y) * p.w + x] * tar[(((long long)b * p.C + c) * p.h + (y + d)) * p.w + x];
        }
      dst[e] = rowok ? -(acc / (float)cpg) : 0.f;
    }
  }
}

// The concat channels of the volume (cv < 2C), all L levels from ONE pass over the source rows: a workgroup owns PSM_RB rows of one (b, cv)
// plane, a lane a 16-byte column segment; the reference half stores one loaded value L times, the target half loads row y + d_l per level
// (the L shifted row sets overlap: L1 / L2 hits) -- the feature maps leave HBM once instead of once per level and XCD, and the stores are
// streaming (the volume is far larger than any cache).  w % 4 == 0.
typedef float psm_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void psm_volume_concat_kernel(const float* __restrict__ ref, const float* __restrict__ tar, float* __restrict__ vol, PsmP p) {
  const int CV = 2 * p.C + p.G;
  const int hb = (p.h + PSM_RB - 1) / PSM_RB;
  long long blk = blockIdx.x;
  const int y0 = (int)(blk % hb) * PSM_RB; blk /= hb;
  const int cv = (int)(blk % (2 * p.C));
  const int b = (int)(blk / (2 * p.C));
  const int nrow = min(PSM_RB, p.h - y0);
  const int n = nrow * p.w;
  const long long plane = (long long)p.h * p.w;
  float* dst0 = vol + (((long long)b * CV + cv) * p.L * p.h + y0) * p.w;
  const bool is_ref = cv < p.C;
  const float* src = (is_ref ? ref + ((long long)b * p.C + cv) * plane : tar + ((long long)b * p.C + (cv - p.C)) * plane) + (long long)y0 * p.w;
  const psm_f4 z = {0.f, 0.f, 0.f, 0.f};
  for (int e = 4 * threadIdx.x; e < n; e += 1024) {
    const int y = y0 + e / p.w;
    psm_f4 v = z;
    if (is_ref) v = *reinterpret_cast<const psm_f4*>(src + e);
#pragma unroll 4
    for (int l = 0; l < p.L; ++l) {
      const int d = p.shift[l];
      const bool rowok = d >= 0 ? (y < p.h - d) : (y >= -d);     // rows the reference writes (psmnet/modules.py:229-246)
      psm_f4 o = z;
      if (rowok) o = is_ref ? v : *reinterpret_cast<const psm_f4*>(src + e + (long long)d * p.w);
      __builtin_nontemporal_store(o, reinterpret_cast<psm_f4*>(dst0 + (long long)l * plane + e));
    }
  }
}

// gradient of the PSMNet volume w.r.t. both feature maps, gather form (one thread per feature element, no atomics):
//   dref[c][y]  = sum_l [row y written at level l] ( dvol[c][l][y]     - corr_l(tar[c][y + d_l]) )
//   dtar[c][y'] = sum_l [y = y' - d_l written]     ( dvol[C + c][l][y] - corr_l(ref[c][y]) ),   corr_l(v) = dvol[2C + g(c)][l][y] * v / cpg
__global__ __launch_bounds__(256) void psm_volume_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ tar,
                                                             const float* __restrict__ dvol, float* __restrict__ dref, float* __restrict__ dtar,
                                                             PsmP p) {
  const int CV = 2 * p.C + p.G;
  const long long plane = (long long)p.h * p.w;
  const long long total = (long long)p.B * p.C * plane;
  const int cpg = p.G > 0 ? p.C / p.G : 1;
  const float inv = 1.f / (float)cpg;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % p.w);
    const int y = (int)((i / p.w) % p.h);
    const int c = (int)((i / plane) % p.C);
    const int b = (int)(i / (plane * p.C));
    const float* dv = dvol + (long long)b * CV * p.L * plane;
    const float* refc = ref + ((long long)b * p.C + c) * plane;
    const float* tarc = tar + ((long long)b * p.C + c) * plane;
    const int g = p.G > 0 ? c / cpg : 0;
    float gr = 0.f, gt = 0.f;
    for (int l = 0; l < p.L; ++l) {
      const int d = p.shift[l];
      const long long lo = ((long long)l * p.h) * p.w + x;
      // reference side: this row itself
      if (d >= 0 ? (y < p.h - d) : (y >= -d)) {
        gr += dv[(long long)c * p.L * plane + lo + (long long)y * p.w];
        if (p.G > 0) gr -= dv[(long long)(2 * p.C + g) * p.L * plane + lo + (long long)y * p.w] * tarc[(long long)(y + d) * p.w + x] * inv;
      }
      // target side: the volume row ys = y - d reads target row y
      const int ys = y - d;
      if (ys >= 0 && ys < p.h && (d >= 0 ? (ys < p.h - d) : (ys >= -d))) {
        gt += dv[(long long)(p.C + c) * p.L * plane + lo + (long long)ys * p.w];
        if (p.G > 0) gt -= dv[(long long)(2 * p.C + g) * p.L * plane + lo + (long long)ys * p.w] * refc[(long long)ys * p.w + x] * inv;
      }
    }
    dref[i] = gr;
    dtar[i] = gt;
  }
}

// StereoNet's difference volume (src/model/stereonet/mainmodel.py:97-112): vol[b,c,l,y,x] = ref[b,c,y,x] - tar[b,c,y+d_l,x] on the rows the
// reference writes (same row rule as the PSMNet volume), 0 elsewhere.  One thread per 4 output columns.
__global__ __launch_bounds__(256) void diff_volume_kernel(const float* __restrict__ ref, const float* __restrict__ tar, float* __restrict__ vol,
                                                          PsmP p) {
  const long long plane = (long long)p.h * p.w;
  const long long total = (long long)p.B * p.C * p.L * plane;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * 1024) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = i + j;
      if (e >= total) break;
      const int x = (int)(e % p.w);
      const int y = (int)((e / p.w) % p.h);
      const int l = (int)((e / plane) % p.L);
      const long long bc = e / (plane * p.L);
      const int d = p.shift[l];
      const bool rowok = d >= 0 ? (y < p.h - d) : (y >= -d);
      vol[e] = rowok ? ref[bc * plane + (long long)y * p.w + x] - tar[bc * plane + (long long)(y + d) * p.w + x] : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void diff_volume_bwd_kernel(const float* __restrict__ dvol, float* __restrict__ dref, float* __restrict__ dtar,
                                                              PsmP p) {
  const long long plane = (long long)p.h * p.w;
  const long long total = (long long)p.B * p.C * plane;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % p.w);
    const int y = (int)((i / p.w) % p.h);
    const long long bc = i / plane;
    const float* dv = dvol + bc * p.L * plane + x;
    float gr = 0.f, gt = 0.f;
    for (int l = 0; l < p.L; ++l) {
      const int d = p.shift[l];
      if (d >= 0 ? (y < p.h - d) : (y >= -d)) gr += dv[((long long)l * p.h + y) * p.w];
      const int ys = y - d;
      if (ys >= 0 && ys < p.h && (d >= 0 ? (ys < p.h - d) : (ys >= -d))) gt -= dv[((long long)l * p.h + ys) * p.w];
    }
    dref[i] = gr;
    dtar[i] = gt;
  }
}

}  // namespace

extern "C" {

int dpf_shift_triple_forward(const float* fea, float* out, const int* iy, const float* wy, const int* ix, const float* wx, int B, int C,
                             int h, int w, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!fea || !out || !iy || !wy || !ix || !wx || B <= 0 || C <= 0 || h <= 0 || w <= 0) return DPF_ERR_INVALID_ARG;
  const long long BC = (long long)B * C, hw = (long long)h * w;
  // the plane index rides in gridDim.y (<= 65535): larger batch x channel counts are launched in slices
  for (long long bc0 = 0; bc0 < BC; bc0 += 21845) {
    const long long n = BC - bc0 < 21845 ? BC - bc0 : 21845;
    hipLaunchKernelGGL(shift_triple_fwd_kernel, dim3(dpf_div_up(h, ST_ROWS), (unsigned)(n * 3)), dim3(256), 0, (hipStream_t)stream,
                       fea + bc0 * hw, out + bc0 * 3 * hw, iy, wy, ix, wx, h, w);
  }
  return dpf_check_launch();
}

int dpf_shift_triple_backward(const float* g, float* dfea, const int* iy, const float* wy, const int* ix, const float* wx, int B, int C,
                              int h, int w, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !dfea || !iy || !wy || !ix || !wx || B <= 0 || C <= 0 || h <= 0 || w <= 0) return DPF_ERR_INVALID_ARG;
  const long long BC = (long long)B * C;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dfea, 0, sizeof(float) * (size_t)BC * h * w, st) != hipSuccess) return DPF_ERR_LAUNCH;
  hipLaunchKernelGGL(shift_triple_bwd_kernel, dim3(dpf_ew_grid(BC * 3 * h * w)), dim3(256), 0, st, g, dfea, iy, wy, ix, wx, BC, h, w);
  return dpf_check_launch();
}

int dpf_shift_triple_backward_gather(const float* g, float* dfea, const int* iy_inv, const float* wy, const int* ix_inv, const float* wx,
                                     int B, int C, int h, int w, void* stream) {
  dpf_clear_error();
  if (!g || !dfea || !iy_inv || !wy || !ix_inv || !wx || B <= 0 || C <= 0 || h <= 0 || w <= 0) return DPF_ERR_INVALID_ARG;
  const long long BC = (long long)B * C, hw = (long long)h * w;
  for (long long bc0 = 0; bc0 < BC; bc0 += 65535) {      // gridDim.y slices
    const long long n = BC - bc0 < 65535 ? BC - bc0 : 65535;
    hipLaunchKernelGGL(shift_triple_bwd_gather_kernel, dim3(dpf_div_up(h, ST_ROWS), (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       g + bc0 * 3 * hw, dfea + bc0 * hw, iy_inv, wy, ix_inv, wx, h, w);
  }
  return dpf_check_launch();
}

int dpf_phase_shift(const float* src, long long src_plane_stride, float* dst, long long dst_plane_stride, const float* mr, const float* hm,
                    float scale, float* tbuf, long long planes, int h, int w, void* stream) {
  dpf_clear_error();
  if (!src || !dst || !mr || !hm || !tbuf || planes <= 0 || planes > 2147483647LL || h <= 0 || w <= 0) return DPF_ERR_INVALID_ARG;
  const size_t lds_a = sizeof(float) * 3 * (size_t)w;
  const size_t lds_b = sizeof(float) * ((size_t)((h + 1) & ~1) * 32 + 2 * (size_t)h + 64);
  if (lds_a > 160 * 1024 || lds_b > 160 * 1024) return DPF_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // the attribute belongs to the (function, device) pair: set once per device, and a failure is an error, not a later launch fault
  static unsigned long long attr_done_mask = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DPF_ERR_LAUNCH;
  if (dev < 0 || dev >= 64 || !((attr_done_mask >> dev) & 1ull)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(phase_colsum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(phase_circ_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return DPF_ERR_LAUNCH;
    if (dev >= 0 && dev < 64) attr_done_mask |= 1ull << dev;
  }
  for (long long p0 = 0; p0 < planes; p0 += 65535) {      // gridDim.y slices
    const long long n = planes - p0 < 65535 ? planes - p0 : 65535;
    hipLaunchKernelGGL(phase_colsum_kernel, dim3((unsigned)n), dim3(256), lds_a, st, src + p0 * src_plane_stride, src_plane_stride, hm, scale,
                       tbuf + p0 * w, h, w);
    hipLaunchKernelGGL(phase_circ_kernel, dim3(dpf_div_up(w, 32), (unsigned)n), dim3(256), lds_b, st, src + p0 * src_plane_stride,
                       src_plane_stride, dst + p0 * dst_plane_stride, dst_plane_stride, mr, tbuf + p0 * w, h, w);
  }
  return dpf_check_launch();
}

int dpf_cv_select_forward(const float* x3, const float* s, float* vol, int B, int C, int h, int w, int CV, int L, int choff,
                          unsigned levels, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x3 || !s || !vol || B <= 0 || C <= 0 || L <= 0 || L > 32 || choff < 0 || choff + C > CV) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(cv_select_fwd_kernel, dim3(dpf_ew_grid((long long)B * C * h * w)), dim3(256), 0, (hipStream_t)stream, x3, s, vol, B,
                     C, h, w, CV, L, choff, levels);
  return dpf_check_launch();
}

int dpf_cv_select_backward(const float* x3, const float* s, const float* dvol, float* dx3, float* ds, int B, int C, int h, int w, int CV,
                           int L, int choff, unsigned levels, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x3 || !s || !dvol || !dx3 || !ds || B <= 0 || C <= 0 || L <= 0 || L > 32 || choff < 0 || choff + C > CV) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(cv_select_bwd_kernel, dim3(dpf_ew_grid((long long)B * C * h * w)), dim3(256), 0, (hipStream_t)stream, x3, s, dvol,
                     dx3, ds, B, C, h, w, CV, L, choff, levels);
  return dpf_check_launch();
}

// shifts: L host ints (int(costrange[l]) truncated toward zero by the caller); groups = 0 -> 'psmnet', > 0 -> 'gwcnet'
int dpf_psm_volume_forward(const float* ref, const float* tar, float* vol, const int* shifts_host, int B, int C, int h, int w, int L,
                           int groups, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!ref || !tar || !vol || !shifts_host || B <= 0 || C <= 0 || L <= 0 || L > 16 || groups < 0 || (groups > 0 && C % groups)) return DPF_ERR_INVALID_ARG;
  PsmP p;
  p.B = B; p.C = C; p.h = h; p.w = w; p.L = L; p.G = groups;
  for (int i = 0; i < 16; ++i) p.shift[i] = i < L ? shifts_host[i] : 0;
  const long long hb = (h + PSM_RB - 1) / PSM_RB;
  const bool fused = (w & 3) == 0 && ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(tar) | reinterpret_cast<uintptr_t>(vol)) & 15) == 0;
  p.cv0 = fused ? 2 * C : 0;                                       // the per-level kernel then only computes the correlation channels
  const long long rows = (long long)B * (2 * C + groups - p.cv0) * L * hb, rows2 = (long long)B * 2 * C * hb;
  if (rows > 0x7fffffffLL || rows2 > 0x7fffffffLL) return DPF_ERR_INVALID_ARG;
  if (fused) hipLaunchKernelGGL(psm_volume_concat_kernel, dim3((unsigned)rows2), dim3(256), 0, (hipStream_t)stream, ref, tar, vol, p);
  if (rows > 0) hipLaunchKernelGGL(psm_volume_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, ref, tar, vol, p);
  return dpf_check_launch();
}

int dpf_psm_volume_backward(const float* ref, const float* tar, const float* dvol, float* dref, float* dtar, const int* shifts_host, int B, int C,
                            int h, int w, int L, int groups, void* stream) {
  dpf_clear_error();
  if (!ref || !tar || !dvol || !dref || !dtar || !shifts_host || B <= 0 || C <= 0 || L <= 0 || L > 16 || groups < 0 || (groups > 0 && C % groups))
    return DPF_ERR_INVALID_ARG;
  PsmP p;
  p.B = B; p.C = C; p.h = h; p.w = w; p.L = L; p.G = groups;
  for (int i = 0; i < 16; ++i) p.shift[i] = i < L ? shifts_host[i] : 0;
  hipLaunchKernelGGL(psm_volume_bwd_kernel, dim3(dpf_ew_grid((long long)B * C * h * w)), dim3(256), 0, (hipStream_t)stream, ref, tar, dvol, dref, dtar,
                     p);
  return dpf_check_launch();
}

// StereoNet difference volume: vol [B, C, L, h, w] = ref - shifted target (stereonet/mainmodel.py:97-112) and its adjoint
int dpf_diff_volume_forward(const float* ref, const float* tar, float* vol, const int* shifts_host, int B, int C, int h, int w, int L,
                            void* stream) {
  dpf_clear_error();
  if (!ref || !tar || !vol || !shifts_host || B <= 0 || C <= 0 || L <= 0 || L > 16) return DPF_ERR_INVALID_ARG;
  PsmP p;
  p.B = B; p.C = C; p.h = h; p.w = w; p.L = L; p.G = 0;
  for (int i = 0; i < 16; ++i) p.shift[i] = i < L ? shifts_host[i] : 0;
  hipLaunchKernelGGL(diff_volume_kernel, dim3(dpf_ew_grid(((long long)B * C * L * h * w + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref, tar,
                     vol, p);
  return dpf_check_launch();
}
int dpf_diff_volume_backward(const float* dvol, float* dref, float* dtar, const int* shifts_host, int B, int C, int h, int w, int L,
                             void* stream) {
  dpf_clear_error();
  if (!dvol || !dref || !dtar || !shifts_host || B <= 0 || C <= 0 || L <= 0 || L > 16) return DPF_ERR_INVALID_ARG;
  PsmP p;
  p.B = B; p.C = C; p.h = h; p.w = w; p.L = L; p.G = 0;
  for (int i = 0; i < 16; ++i) p.shift[i] = i < L ? shifts_host[i] : 0;
  hipLaunchKernelGGL(diff_volume_bwd_kernel, dim3(dpf_ew_grid((long long)B * C * h * w)), dim3(256), 0, (hipStream_t)stream, dvol, dref, dtar, p);
  return dpf_check_launch();
}

}  // extern "C"
