// Convolutions with a handful of output channels (K <= 4): the 32 -> 1 cost heads of the aggregation stack
// (reference: src/model/stereodpnet/modules.py:286-296 `classif*[2]`) and the last 32 -> 3 normal conv
// (normal_module.py:65).  A 32-row MFMA tile would be 3-10 % utilised here; the work is HBM-bound instead
// (read C channels once, write K), so these are direct kernels: forward = one thread per output voxel with the weights
// broadcast from LDS; weight gradient = one thread per (channel, tap) pair marching over an LDS-staged voxel tile.
#include "dpf_common.h"

namespace {

constexpr int MAXK = 4;
constexpr int TW = 32, TH = 8;   // wgrad position tile

struct SkP {
  int N, C, K;
  int ID, IH, IW, OD, OH, OW;
  int kd, kh, kw, T;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
};

// forward, stride-1 W: one thread owns XB = 4 consecutive outputs along W of one (n, od, oh) row; for every (channel, kd, kh)
// it loads the XB + (kw-1)*dw input values once and reuses them for the kw taps (weights are wave-uniform -> scalar loads).
constexpr int XB = 4;
constexpr int MAXSPAN = XB + 2;       // kw <= 3, dw == 1

__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, SkP p) {
  const long long oplane = (long long)p.OH * p.OW, ovol = oplane * p.OD;
  const long long ivol = (long long)p.ID * p.IH * p.IW;
  const int wq = (p.OW + XB - 1) / XB;
  const long long total = (long long)p.N * p.OD * p.OH * wq;
  const int span = XB + (p.kw - 1);                 // stride 1, dilation 1 along W (checked by the host)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow0 = (int)(i % wq) * XB;
    const int oh = (int)((i / wq) % p.OH);
    const int od = (int)((i / ((long long)wq * p.OH)) % p.OD);
    const int n = (int)(i / ((long long)wq * p.OH * p.OD));
    float acc[MAXK][XB];
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
#pragma unroll
      for (int j = 0; j < XB; ++j) acc[k][j] = (bias && k < p.K) ? bias[k] : 0.f;
    const float* xn = x + (long long)n * p.C * ivol;
    const int iw0 = ow0 - p.pw;
    for (int c = 0; c < p.C; ++c) {
      for (int a = 0; a < p.kd; ++a) {
        const int id = od * p.sd - p.pd + a * p.dd;
        if (id < 0 || id >= p.ID) continue;
        for (int b = 0; b < p.kh; ++b) {
          const int ih = oh * p.sh - p.ph + b * p.dh;
          if (ih < 0 || ih >= p.IH) continue;
          const float* row = xn + (long long)c * ivol + ((long long)id * p.IH + ih) * p.IW;
          float v[MAXSPAN];
#pragma unroll
          for (int u = 0; u < MAXSPAN; ++u) {
            const int iw = iw0 + u;
            v[u] = (u < span && iw >= 0 && iw < p.IW) ? row[iw] : 0.f;
          }
          const int tb = (a * p.kh + b) * p.kw;
#pragma unroll
          for (int k = 0; k < MAXK; ++k) {
            if (k < p.K) {
              const float* wk = w + ((long long)k * p.C + c) * p.T + tb;
              for (int c2 = 0; c2 < p.kw; ++c2) {
                const float wv = wk[c2];
#pragma unroll
                for (int j = 0; j < XB; ++j) {
                  const float xv = c2 == 0 ? v[j] : (c2 == 1 ? v[j + 1] : v[j + 2]);
                  acc[k][j] = fmaf(wv, xv, acc[k][j]);
                }
              }
            }
          }
        }
      }
    }
    const long long obase = ((long long)od * p.OH + oh) * p.OW + ow0;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < p.K)
#pragma unroll
        for (int j = 0; j < XB; ++j)
          if (ow0 + j < p.OW) out[((long long)n * p.K + k) * ovol + obase + j] = acc[k][j];
  }
}

// grid = cchunks * nblk; block: channels [c0, c0+CCH), tiles pchunk, pchunk+nblk, ...
// thread = (channel, kd, kh) triple x row group; it marches along W with a 3-deep sliding window of g so that each
// staged x value (one LDS read) feeds the kw = 3 taps: 2 LDS reads per 3*K FMAs.
template <int CCH>
__global__ __launch_bounds__(256) void smallk_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw, SkP p,
                                                           int nblk, int tilesH, int tilesW, long long ntiles) {
  extern __shared__ float smem[];
  const int ext_d = (p.kd - 1) * p.dd + 1;
  const int ext_h = (TH - 1) * p.sh + (p.kh - 1) * p.dh + 1;
  const int ext_w = TW + p.kw - 1;                     // stride 1, dilation 1 along W (host-checked)
  const int chanStride = ext_d * ext_h * ext_w;
  float* s_x = smem;                       // [CCH][chanStride]
  float* s_g = s_x + CCH * chanStride;     // [K][TH][TW]
  int* s_rowoff = (int*)(s_g + MAXK * TH * TW);      // staged-row source offsets (no divisions in the tile loop)
  int* s_rowpr = s_rowoff + CCH * ext_d * ext_h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int cchunk = blockIdx.x / nblk, pchunk = blockIdx.x % nblk;
  const int c0 = cchunk * CCH;
  const int ncc = min(CCH, p.C - c0);
  {
    const long long xv_ = (long long)p.ID * p.IH * p.IW;
    const int rpc = ext_d * ext_h;
    for (int rowid = tid; rowid < ncc * rpc; rowid += 256) {
      const int c2 = rowid / rpc;
      const int rem = rowid - c2 * rpc;
      const int pl = rem / ext_h, rr = rem - pl * ext_h;
      s_rowoff[rowid] = (int)((long long)c2 * xv_ + ((long long)pl * p.IH + rr) * p.IW);
      s_rowpr[rowid] = (pl << 16) | rr;
    }
  }
  const int ntrip = ncc * p.kd * p.kh;
  int nrg = 256 / ntrip;                   // row groups
  nrg = nrg >= 8 ? 8 : (nrg >= 4 ? 4 : (nrg >= 2 ? 2 : 1));
  const bool active = tid < ntrip * nrg;
  const int trip = active ? tid % ntrip : 0;
  const int rg = active ? tid / ntrip : 0;
  const int cc = trip / (p.kd * p.kh);
  const int tdh = trip - cc * (p.kd * p.kh);
  const int td_ = tdh / p.kh, th_ = tdh - td_ * p.kh;
  const int base = cc * chanStride + (td_ * p.dd * ext_h + th_ * p.dh) * ext_w;
  float acc[MAXK][3];
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[k][j] = 0.f;
  const long long xvol = (long long)p.ID * p.IH * p.IW, gvol = (long long)p.OD * p.OH * p.OW;
  const int rows_per_chan = ext_d * ext_h;
  for (long long tile = pchunk; tile < ntiles; tile += nblk) {
    long long b = tile;
    const int tw = (int)(b % tilesW); b /= tilesW;
    const int th = (int)(b % tilesH); b /= tilesH;
    const int qd = (int)(b % p.OD);
    const int n = (int)(b / p.OD);
    const int q0h = th * TH, q0w = tw * TW;
    const int i0d = qd * p.sd - p.pd, i0h = q0h * p.sh - p.ph, i0w = q0w - p.pw;
    __syncthreads();
    const float* xbase = x + ((long long)n * p.C + c0) * xvol + ((long long)i0d * p.IH + i0h) * p.IW;
    constexpr int SU = 8;
    const int xrows = ncc * rows_per_chan;
    for (int r0 = wave_u * SU; r0 < xrows; r0 += 4 * SU) {
      float v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int rowid = r0 + u;
        const int rsafe = rowid < xrows ? rowid : 0;
        const int pr = s_rowpr[rsafe];
        const int id = i0d + (pr >> 16), ih = i0h + (pr & 0xffff);
        const int iw = i0w + lane;
        const bool ok = rowid < xrows && lane < ext_w && id >= 0 && id < p.ID && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
        v[u] = ok ? xbase[s_rowoff[rsafe] + iw] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int rowid = r0 + u;
        if (rowid < xrows && lane < ext_w) s_x[rowid * ext_w + lane] = v[u];
      }
    }
    for (int i = tid; i < p.K * TH * TW; i += 256) {
      const int k = i / (TH * TW), r = (i / TW) % TH, cx = i % TW;
      const int qh = q0h + r, qw = q0w + cx;
      s_g[i] = (qh < p.OH && qw < p.OW) ? g[((long long)n * p.K + k) * gvol + ((long long)qd * p.OH + qh) * p.OW + qw] : 0.f;
    }
    __syncthreads();
    if (active) {
      for (int r = rg; r < TH; r += nrg) {
        const float* xr = s_x + base + (r * p.sh) * ext_w;
        const float* gr = s_g + r * TW;
        float g0[MAXK], g1[MAXK], g2[MAXK];      // g[xi], g[xi-1], g[xi-2]
#pragma unroll
        for (int k = 0; k < MAXK; ++k) g0[k] = g1[k] = g2[k] = 0.f;
#pragma unroll 2
        for (int xi = 0; xi < TW + 2; ++xi) {
          const float xv = xi < ext_w ? xr[xi] : 0.f;
#pragma unroll
          for (int k = 0; k < MAXK; ++k) {
            g2[k] = g1[k];
            g1[k] = g0[k];
            g0[k] = (k < p.K && xi < TW) ? gr[k * TH * TW + xi] : 0.f;
            // x column xi pairs with output column xi - tw for tap tw
            acc[k][0] = fmaf(g0[k], xv, acc[k][0]);
            acc[k][1] = fmaf(g1[k], xv, acc[k][1]);
            acc[k][2] = fmaf(g2[k], xv, acc[k][2]);
          }
        }
      }
    }
  }
  if (active) {
    const int tbase = (td_ * p.kh + th_) * p.kw;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < p.K)
        for (int j = 0; j < p.kw; ++j) atomicAdd(&dw[((long long)k * p.C + c0 + cc) * p.T + tbase + j], acc[k][j]);
  }
}

// Forward specialised for 3x3 (x KD) stride-1 / dilation-1 windows and KK output channels: all loops unrolled, the 27 * KK weights
// of a channel are wave-uniform scalar loads.  A thread owns SKR consecutive output rows x XB columns: per (channel, depth tap) it
// loads the SKR + 2 input rows once (XB + 2 values each) and every row feeds the three row taps of up to three outputs -- 1.5 row
// loads per output row instead of 3 (the kernel is bound by cache bandwidth: 32 channels x 27 taps re-read the same lines).
constexpr int SKR = 4;
template <int KK, int KD, bool VEC, int SKRT = SKR>
__global__ __launch_bounds__(256) void smallk_fwd3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, SkP p) {
  const long long oplane = (long long)p.OH * p.OW, ovol = oplane * p.OD;
  const long long iplane = (long long)p.IH * p.IW, ivol = iplane * p.ID;
  const int wq = (p.OW + XB - 1) / XB;
  const int hq = (p.OH + SKRT - 1) / SKRT;
  const long long total = (long long)p.N * p.OD * hq * wq;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ow0 = (int)(i % wq) * XB;
    const int oh0 = (int)((i / wq) % hq) * SKRT;
    const int od = (int)((i / ((long long)wq * hq)) % p.OD);
    const int n = (int)(i / ((long long)wq * hq * p.OD));
    float acc[KK][SKRT][XB];
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
      for (int r = 0; r < SKRT; ++r)
#pragma unroll
        for (int j = 0; j < XB; ++j) acc[k][r][j] = bias ? bias[k] : 0.f;
    const float* xn = x + (long long)n * p.C * ivol;
    const int iw0 = ow0 - p.pw;
    bool cok[XB + 2];
#pragma unroll
    for (int u = 0; u < XB + 2; ++u) cok[u] = iw0 + u >= 0 && iw0 + u < p.IW;
    for (int c = 0; c < p.C; ++c) {
      const float* xc = xn + (long long)c * ivol;
      const float* wc = w + (long long)c * (KD * 9);
      // all row loads of the channel first, into their own registers: with one window array reused per row the compiler waits for
      // every row before it issues the next one ((SKRT + 2) * KD dependent round trips to L2 / HBM per channel, ~13 us)
      float vv[KD][SKRT + 2][XB + 2];
#pragma unroll
      for (int a = 0; a < KD; ++a) {
        const int id = od - p.pd + a * p.dd;
        const bool dok = id >= 0 && id < p.ID;
#pragma unroll
        for (int rr = 0; rr < SKRT + 2; ++rr) {                 // input row ih = oh0 - ph + rr feeds outputs oh0 + rr - b, b = 0..2
          const int ih = oh0 - p.ph + rr;
          const bool rok = dok && ih >= 0 && ih < p.IH;
          float (&v)[XB + 2] = vv[a][rr];
          if (VEC) {
            // pw == 1 and IW % 4 == 0: the four centre values are one aligned 16-byte load (consecutive lanes: 1 KiB contiguous).
            // Loads are UNCONDITIONAL from clamped (always valid) addresses and zeroed by selects afterwards: a conditional load is a
            // branch, and behind a branch the compiler retires every row before it issues the next one.
            const int idc = id < 0 ? 0 : (id >= p.ID ? p.ID - 1 : id), ihc = ih < 0 ? 0 : (ih >= p.IH ? p.IH - 1 : ih);
            const float* rowc = xc + (long long)idc * iplane + (long long)ihc * p.IW + iw0;
            const float4 mid = *reinterpret_cast<const float4*>(rowc + 1);
            const float e0 = rowc[cok[0] ? 0 : 1], e5 = rowc[cok[5] ? 5 : 4];
            v[0] = (rok && cok[0]) ? e0 : 0.f;
            v[1] = rok ? mid.x : 0.f; v[2] = rok ? mid.y : 0.f; v[3] = rok ? mid.z : 0.f; v[4] = rok ? mid.w : 0.f;
            v[5] = (rok && cok[5]) ? e5 : 0.f;
          } else {
            const float* row = xc + (long long)id * iplane + (long long)ih * p.IW + iw0;
#pragma unroll
            for (int u = 0; u < XB + 2; ++u) v[u] = (rok && cok[u]) ? row[u] : 0.f;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < KD; ++a) {
#pragma unroll
        for (int rr = 0; rr < SKRT + 2; ++rr) {
          const float (&v)[XB + 2] = vv[a][rr];
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const int r = rr - b;                               // compile-time after unrolling
            if (r < 0 || r >= SKRT) continue;
#pragma unroll
            for (int k = 0; k < KK; ++k) {
              const float* wk = wc + (long long)k * p.C * (KD * 9) + (a * 3 + b) * 3;
              const float w0 = wk[0], w1 = wk[1], w2 = wk[2];
#pragma unroll
              for (int j = 0; j < XB; ++j) acc[k][r][j] = fmaf(w0, v[j], fmaf(w1, v[j + 1], fmaf(w2, v[j + 2], acc[k][r][j])));
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < SKRT; ++r) {
      if (oh0 + r >= p.OH) break;
      const long long obase = ((long long)od * p.OH + oh0 + r) * p.OW + ow0;
#pragma unroll
      for (int k = 0; k < KK; ++k)
#pragma unroll
        for (int j = 0; j < XB; ++j)
          if (ow0 + j < p.OW) out[((long long)n * p.K + k) * ovol + obase + j] = acc[k][r][j];
    }
  }
}

// Data gradient of a conv with K <= 4 output channels (3 x 3 (x KD) window, stride 1, dilation 1): dx[n,c,i] = sum_{k,t} w[k,c,t] g[n,k,i+p-t].
// The implicit-GEMM kernels run this shape with one real reduction channel per MFMA step (11 TFLOP/s = 0.48 ms for the 1 -> 32
// channel cost heads at 8 x 256 x 384); here a thread owns 4 consecutive W positions, loads the K x KD x 3 rows of g it needs ONCE
// (an aligned float4 + 2 edge values per row) and produces all C input channels from registers: 27 K wave-uniform (scalar) weight
// loads and one float4 store per channel -- the kernel is a 402 MB stream of stores.
template <int KK, int KD>
__global__ __launch_bounds__(256) void smallk_dgrad3_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx, SkP p) {
  const int wq = p.IW / 4;
  const long long total = (long long)p.N * p.ID * p.IH * wq;
  const long long oplane = (long long)p.OH * p.OW, ovol = oplane * p.OD;
  const long long iplane = (long long)p.IH * p.IW, ivol = iplane * p.ID;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int iw0 = (int)(i % wq) * 4;
    const int ih = (int)((i / wq) % p.IH);
    const int id = (int)((i / ((long long)wq * p.IH)) % p.ID);
    const int n = (int)(i / ((long long)wq * p.IH * p.ID));
    // g window: v[k][a][b][u] = g[n][k][id + pd - a][ih + ph - b][iw0 + pw - 2 + u], u = 0..5 (tap cc of position j reads u = j + 2 - cc)
    float v[KK][KD][3][6];
    const int ow0 = iw0 + p.pw - 2;
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
      for (int a = 0; a < KD; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int od = id + p.pd - a, oh = ih + p.ph - b;
          const bool rok = od >= 0 && od < p.OD && oh >= 0 && oh < p.OH;
          const float* row = g + ((long long)n * p.K + k) * ovol + (long long)od * oplane + (long long)oh * p.OW + ow0;
          if (p.pw == 1 && (p.OW & 3) == 0) {       // ow0 + 1 = iw0: aligned float4 in the middle
            const float4 mid = rok ? *reinterpret_cast<const float4*>(row + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[k][a][b][0] = (rok && ow0 >= 0) ? row[0] : 0.f;
            v[k][a][b][1] = mid.x; v[k][a][b][2] = mid.y; v[k][a][b][3] = mid.z; v[k][a][b][4] = mid.w;
            v[k][a][b][5] = (rok && ow0 + 5 < p.OW) ? row[5] : 0.f;
          } else {
#pragma unroll
            for (int u = 0; u < 6; ++u) v[k][a][b][u] = (rok && ow0 + u >= 0 && ow0 + u < p.OW) ? row[u] : 0.f;
          }
        }
    float* dst = dx + (long long)n * p.C * ivol + (long long)id * iplane + (long long)ih * p.IW + iw0;
    for (int c = 0; c < p.C; ++c) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        const float* wk = w + ((long long)k * p.C + c) * (KD * 9);          // wave-uniform: scalar loads
#pragma unroll
        for (int a = 0; a < KD; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const float w0 = wk[(a * 3 + b) * 3], w1 = wk[(a * 3 + b) * 3 + 1], w2 = wk[(a * 3 + b) * 3 + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = fmaf(w0, v[k][a][b][j + 2], fmaf(w1, v[k][a][b][j + 1], fmaf(w2, v[k][a][b][j], acc[j])));
          }
      }
      *reinterpret_cast<float4*>(dst + (long long)c * ivol) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// Weight gradient, 3x3 (x KD) stride-1 windows: no LDS.  One wave owns (n, od, channel, 64-column strip, row range) and marches
// down the rows with lanes along W: per step it loads the newly entering input row of each kernel plane in its three column
// shifts (9 coalesced loads for KD = 3, served by L1 after the first), keeps the other two rows of the window in registers, and
// feeds KD*9 FMAs per output channel from them.  The KD*9*K per-lane partial sums are reduced across the wave once, at the end.
template <int KK, int KD>
__global__ __launch_bounds__(256) void smallk_wgrad_rows_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw,
                                                                SkP p, int segs, int rsplit, int rows_per) {
  const int lane = threadIdx.x & 63;
  long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = (int)(id % p.C); id /= p.C;           // the 4 waves of a block share g rows (and x addresses modulo the channel)
  const int seg = (int)(id % segs); id /= segs;
  const int rs = (int)(id % rsplit); id /= rsplit;
  const int od = (int)(id % p.OD);
  const long long n = id / p.OD;
  if (n >= p.N) return;
  const int ow = seg * 64 + lane;
  const int oh0 = rs * rows_per, oh1 = min(p.OH, oh0 + rows_per);
  const long long iplane = (long long)p.IH * p.IW, ivol = iplane * p.ID, ovol = (long long)p.OD * p.OH * p.OW;
  const float* xc = x + ((long long)n * p.C + c) * ivol;
  const float* gn = g + (long long)n * p.K * ovol + (long long)od * p.OH * p.OW;
  // per kernel plane: base pointer (or null when the plane is outside the volume)
  const float* xpl[KD];
#pragma unroll
  for (int a = 0; a < KD; ++a) {
    const int idp = od - p.pd + a * p.dd;
    xpl[a] = (idp >= 0 && idp < p.ID) ? xc + (long long)idp * iplane : nullptr;
  }
  int iw[3];
  bool cok[3];
#pragma unroll
  for (int sft = 0; sft < 3; ++sft) {
    iw[sft] = ow - p.pw + sft;
    cok[sft] = iw[sft] >= 0 && iw[sft] < p.IW;
  }
  auto load_row = [&](int ih, float (&dst)[KD][3]) {
    const bool rok = ih >= 0 && ih < p.IH;
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int sft = 0; sft < 3; ++sft) dst[a][sft] = (rok && cok[sft] && xpl[a]) ? xpl[a][(long long)ih * p.IW + iw[sft]] : 0.f;
  };
  float acc[KK][KD][3][3];
#pragma unroll
  for (int k = 0; k < KK; ++k)
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) acc[k][a][b][sft] = 0.f;
  float r0[KD][3], r1[KD][3], r2[KD][3];
  load_row(oh0 - p.ph, r0);
  load_row(oh0 - p.ph + 1, r1);
  const bool gok = ow < p.OW;
  for (int oh = oh0; oh < oh1; ++oh) {
    load_row(oh - p.ph + 2, r2);
    float gv[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) gv[k] = gok ? gn[(long long)k * ovol + (long long)oh * p.OW + ow] : 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
      for (int a = 0; a < KD; ++a)
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) {
          acc[k][a][0][sft] = fmaf(gv[k], r0[a][sft], acc[k][a][0][sft]);
          acc[k][a][1][sft] = fmaf(gv[k], r1[a][sft], acc[k][a][1][sft]);
          acc[k][a][2][sft] = fmaf(gv[k], r2[a][sft], acc[k][a][2][sft]);
        }
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int sft = 0; sft < 3; ++sft) {
        r0[a][sft] = r1[a][sft];
        r1[a][sft] = r2[a][sft];
      }
  }
#pragma unroll
  for (int k = 0; k < KK; ++k)
#pragma unroll
    for (int a = 0; a < KD; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) {
          float v = acc[k][a][b][sft];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0 && v != 0.f) atomicAdd(&dw[((long long)k * p.C + c) * p.T + (a * 3 + b) * 3 + sft], v);
        }
}

int fill(SkP& p, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd,
         int dh, int dw) {
  if (N <= 0 || C <= 0 || K <= 0 || K > MAXK) return DPF_ERR_UNSUPPORTED;
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.T = kd * kh * kw;
  p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.OD = (ID + 2 * pd - (dd * (kd - 1) + 1)) / sd + 1;
  p.OH = (IH + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  p.OW = (IW + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  if (p.OD <= 0 || p.OH <= 0 || p.OW <= 0 || p.T > 27) return DPF_ERR_INVALID_ARG;
  return DPF_OK;
}

}  // namespace

extern "C" {

// same tensor conventions as dpf_conv_forward; K <= 4, stride 1 / dilation 1 / kw <= 3 along W
int dpf_conv_smallk_forward(const float* x, const float* w, const float* bias, float* out, int N, int C, int ID, int IH, int IW, int K, int kd,
                            int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, void* stream) {
  dpf_clear_error();
  if (!x || !w || !out) return DPF_ERR_INVALID_ARG;
  SkP p{};
  int rc = fill(p, N, C, ID, IH, IW, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw);
  if (rc != DPF_OK) return rc;
  if (sw != 1 || dw != 1 || kw > 3) return DPF_ERR_UNSUPPORTED;
  const long long total = (long long)N * p.OD * p.OH * ((p.OW + XB - 1) / XB);
  if (kh == 3 && kw == 3 && (kd == 1 || kd == 3) && sd == 1 && sh == 1 && dh == 1) {
    const dim3 grid(dpf_ew_grid((long long)N * p.OD * ((p.OH + SKR - 1) / SKR) * ((p.OW + XB - 1) / XB)));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = pw == 1 && (IW & 3) == 0 && (p.OW & 3) == 0 && XB == 4;
    // rows per thread: 4 (default; 0.30 ms per 32 -> 1 cost head at 8 x 256 x 384 once the loads are unconditional) or 2 (0.37 ms)
    constexpr int rows_over = 4;
    if (K == 1 && kd == 3 && vec && rows_over == 2) {
      const dim3 grid2(dpf_ew_grid((long long)N * p.OD * ((p.OH + 1) / 2) * ((p.OW + XB - 1) / XB)));
      hipLaunchKernelGGL((smallk_fwd3_kernel<1, 3, true, 2>), grid2, dim3(256), 0, st, x, w, bias, out, p);
      return dpf_check_launch();
    }
#define DPF_SKF(KKv, KDv)                                                                                         \
  {                                                                                                               \
    if (vec) hipLaunchKernelGGL((smallk_fwd3_kernel<KKv, KDv, true>), grid, dim3(256), 0, st, x, w, bias, out, p); \
    else hipLaunchKernelGGL((smallk_fwd3_kernel<KKv, KDv, false>), grid, dim3(256), 0, st, x, w, bias, out, p);   \
  }
    if (kd == 1) {
      switch (K) { case 1: DPF_SKF(1, 1); break; case 2: DPF_SKF(2, 1); break; case 3: DPF_SKF(3, 1); break; default: DPF_SKF(4, 1); break; }
    } else {
      switch (K) { case 1: DPF_SKF(1, 3); break; case 2: DPF_SKF(2, 3); break; case 3: DPF_SKF(3, 3); break; default: DPF_SKF(4, 3); break; }
    }
#undef DPF_SKF
    return dpf_check_launch();
  }
  hipLaunchKernelGGL(smallk_fwd_kernel, dim3(dpf_ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, p);
  return dpf_check_launch();
}

// data gradient of dpf_conv_smallk_forward's shapes: g [N,K,OD,OH,OW] (K <= 4) -> dx [N,C,ID,IH,IW]; 3 x 3 (x 1 or 3) windows, stride 1,
// dilation 1, IW % 4 == 0 and 16-byte aligned dx; DPF_ERR_UNSUPPORTED otherwise (the caller then uses dpf_conv_transpose)
int dpf_conv_smallk_dgrad(const float* g, const float* w, float* dx, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int pd,
                          int ph, int pw, void* stream) {
  dpf_clear_error();
  if (!g || !w || !dx) return DPF_ERR_INVALID_ARG;
  SkP p{};
  int rc = fill(p, N, C, ID, IH, IW, K, kd, kh, kw, 1, 1, 1, pd, ph, pw, 1, 1, 1);
  if (rc != DPF_OK) return rc;
  if (kh != 3 || kw != 3 || (kd != 1 && kd != 3) || (IW & 3) || (reinterpret_cast<uintptr_t>(dx) & 15) || (reinterpret_cast<uintptr_t>(g) & 15) ||
      K * kd > 6)
    return DPF_ERR_UNSUPPORTED;
  const long long total = (long long)N * ID * IH * (IW / 4);
  const dim3 grid(dpf_ew_grid(total));
  hipStream_t st = (hipStream_t)stream;
#define DPF_SKD(KKv, KDv) hipLaunchKernelGGL((smallk_dgrad3_kernel<KKv, KDv>), grid, dim3(256), 0, st, g, w, dx, p)
  if (kd == 3) {
    if (K == 1) DPF_SKD(1, 3); else DPF_SKD(2, 3);
  } else {
    switch (K) { case 1: DPF_SKD(1, 1); break; case 2: DPF_SKD(2, 1); break; case 3: DPF_SKD(3, 1); break; default: DPF_SKD(4, 1); break; }
  }
#undef DPF_SKD
  return dpf_check_launch();
}

// dw[K][C][T] += sum g[n,k,q] * x[n,c,q*s - p + t*dil]   (g [N,K,OD,OH,OW], x [N,C,ID,IH,IW])
int dpf_conv_smallk_wgrad(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int kd, int kh, int kw, int sd,
                          int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_, void* stream) {
  dpf_clear_error();
  if (!g || !x || !dw) return DPF_ERR_INVALID_ARG;
  SkP p{};
  int rc = fill(p, N, C, ID, IH, IW, K, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw_);
  if (rc != DPF_OK) return rc;
  if (kh == 3 && kw == 3 && (kd == 1 || kd == 3) && sd == 1 && sh == 1 && sw == 1 && dh == 1 && dw_ == 1) {
    // register-window kernel; split the rows of short strips so that the grid still fills the chip
    const int segs = dpf_div_up(p.OW, 64);
    const long long strips = (long long)N * p.OD * segs * C;
    int rsplit = 1;
    while (strips * rsplit < 8192 && rsplit * 32 < p.OH) rsplit *= 2;
    const int rows_per = dpf_div_up(p.OH, rsplit);
    const long long waves = strips * rsplit;
    const dim3 grid((unsigned)dpf_div_up(waves, 4));
    hipStream_t st = (hipStream_t)stream;
#define DPF_SKW(KKv, KDv) hipLaunchKernelGGL((smallk_wgrad_rows_kernel<KKv, KDv>), grid, dim3(256), 0, st, g, x, dw, p, segs, rsplit, rows_per)
    if (kd == 1) {
      switch (K) { case 1: DPF_SKW(1, 1); break; case 2: DPF_SKW(2, 1); break; case 3: DPF_SKW(3, 1); break; default: DPF_SKW(4, 1); break; }
    } else {
      switch (K) { case 1: DPF_SKW(1, 3); break; case 2: DPF_SKW(2, 3); break; case 3: DPF_SKW(3, 3); break; default: DPF_SKW(4, 3); break; }
    }
#undef DPF_SKW
    return dpf_check_launch();
  }
  constexpr int CCH = 16;
  if (sw != 1 || dw_ != 1 || kw > 3 || CCH * kd * kh > 256) return DPF_ERR_UNSUPPORTED;
  const int ext_d = (kd - 1) * dd + 1, ext_h = (TH - 1) * sh + (kh - 1) * dh + 1, ext_w = TW + kw - 1;
  const size_t lds = sizeof(float) * ((size_t)CCH * ext_d * ext_h * ext_w + (size_t)MAXK * TH * TW + 2 * (size_t)CCH * ext_d * ext_h);
  if (lds > 150 * 1024) return DPF_ERR_UNSUPPORTED;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)smallk_wgrad_kernel<CCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return DPF_ERR_LAUNCH;
  const int tilesH = dpf_div_up(p.OH, TH), tilesW = dpf_div_up(p.OW, TW);
  const long long ntiles = (long long)N * p.OD * tilesH * tilesW;
  const int cchunks = dpf_div_up(C, CCH);
  long long nblk = 1024 / cchunks;
  if (nblk < 1) nblk = 1;
  if (nblk > ntiles) nblk = ntiles;
  hipLaunchKernelGGL((smallk_wgrad_kernel<CCH>), dim3((unsigned)(cchunks * nblk)), dim3(256), lds, (hipStream_t)stream, g, x, dw, p, (int)nblk,
                     tilesH, tilesW, ntiles);
  return dpf_check_launch();
}

}  // extern "C"
