// Dense N-d convolution family (2-D handled as depth 1) as implicit GEMM on the fp32 matrix cores.
//
// Replaces the cuDNN calls behind nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d on the StereoDPNet
// path (reference call sites: src/module/asm/basics.py:17-36, src/model/stereodpnet/modules.py:
// 26-32,64-69,88-91,208-227,271-296, normal_module.py:14-19, dcn3d/modules/deform_conv.py:310-315).
//
//   dpf_conv_forward   out[n,k,o] = bias[k] + sum_{c,t} W[k,c,t] * x[n,c, o*s - p + t*dil]
//   dpf_conv_transpose out[n,k,o] = bias[k] + sum_{c,t} W'[c,k,t] * x[n,c, (o + p - t*dil)/s]   (exact division only)
//                      (= data gradient of the forward conv, and nn.ConvTranspose3d's forward)
//   dpf_conv_wgrad     dW[k,c,t] += sum_{n,q} g[n,k,q] * x[n,c, q*s - p + t*dil]
//
// Mapping (MI355X-first, not a cuDNN/CUDA tiling): v_mfma_f32_32x32x2_f32, exact f32 (bitwise an
// fmaf chain) at the f32 vector peak.  D[row = out channel][col = position]: the weight fragment is
// the A operand (lane&31 = channel), the input patch the B operand (lane&31 = 32 consecutive W
// positions, so every LDS read and every store of a wave half is one contiguous 128-B segment).
// One workgroup (4 waves) owns ALL output channels of an 8x32 position tile of one (n, depth) plane,
// so the input patch (with halo) is staged in LDS once per channel chunk and never re-read for
// another channel tile.  Strided transposed convolutions are decomposed into s^3 parity classes,
// each a dense stride-1 gather over the valid tap subset (no zero-insertion, no wasted MACs).
#include "dpf_common.h"
#include "dpf_repack.h"
#include "conv_internal.h"
#include <cstdlib>

namespace {

constexpr int TW = 32;       // positions along W per tile (= MFMA N)
constexpr int MAXT = 27;

struct ConvP {
  int N, C, K;
  int ID, IH, IW;   // x dims
  int OD, OH, OW;   // out dims
  int kd, kh, kw;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int transposed;
  int QD, tilesH, tilesW, ncls;   // grid decomposition (class-0 sizes)
  int Ktot, k0;                   // out tensor has Ktot channels; this launch writes [k0, k0 + K)
  int chanStrideMax;              // floats per channel of the LDS input tile (host worst case)
  int ntmax;                      // max valid taps per class
  int maxrows;                    // LDS input-tile rows per channel chunk (host worst case)
};

struct DimTap {
  int emin, emax;
};

// per-dimension tap geometry: input coordinate = q*sx + e(t)
__device__ __forceinline__ bool tap_e(int t, int k, int s, int p, int dil, int r, int transposed, int& e) {
  if (!transposed) {
    e = t * dil - p;
    return true;
  }
  const int num = r + p - t * dil;
  int m = num % s;
  if (m < 0) m += s;
  if (m != 0) return false;
  e = (num - m) / s;   // exact
  return true;
}

__device__ __forceinline__ void dim_range(int k, int s, int p, int dil, int r, int transposed, int& emin, int& emax, int& nvalid) {
  emin = 1 << 30;
  emax = -(1 << 30);
  nvalid = 0;
  for (int t = 0; t < k; ++t) {
    int e;
    if (tap_e(t, k, s, p, dil, r, transposed, e)) {
      emin = e < emin ? e : emin;
      emax = e > emax ? e : emax;
      ++nvalid;
    }
  }
}

// NT rows of 32 positions per wave; tile = (4*NT) x 32 positions.  The 4-channel-chunk variants with <= 64 output channels fit
// 128 registers without spilling, so they ask for 4 waves/SIMD (4 resident workgroups per CU hide the staging phases).
template <int MT, int CC, int NT>
__global__ __launch_bounds__(256, (CC == 4 && MT <= 2) ? 4 : 2) void conv_igemm_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, float* __restrict__ out, ConvP p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int KT = 32 * MT;
  constexpr int TH = 4 * NT;
  // LDS carve: [input tile CC*chanStrideMax][tapoff MAXT][tapw MAXT][nv][row tables]
  float* s_in = smem;
  int* s_tapoff = (int*)(s_in + CC * p.chanStrideMax);
  int* s_tapw = s_tapoff + MAXT;
  int* s_nv = s_tapw + MAXT;
  int* s_rowoff = s_nv + 4;             // per staged row: source offset relative to (n, c0, i0d, i0h) -- no divisions in the loop
  int* s_rowpr = s_rowoff + p.maxrows;  // (plane << 16) | row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hh = lane >> 5;

  int b = blockIdx.x;
  const int tw = b % p.tilesW; b /= p.tilesW;
  const int th = b % p.tilesH; b /= p.tilesH;
  const int qd = b % p.QD; b /= p.QD;
  const int n = b % p.N;
  const int cls = b / p.N;

  int rd = 0, rh = 0, rw = 0;
  int sxd = p.sd, sxh = p.sh, sxw = p.sw;   // input step per output step
  int sod = 1, soh = 1, sow = 1;            // output step per q step
  if (p.transposed) {
    rw = cls % p.sw;
    rh = (cls / p.sw) % p.sh;
    rd = cls / (p.sw * p.sh);
    sxd = sxh = sxw = 1;
    sod = p.sd; soh = p.sh; sow = p.sw;
  }
  // q-grid of this class
  const int Qd = (p.OD - rd + sod - 1) / sod;
  const int Qh = (p.OH - rh + soh - 1) / soh;
  const int Qw = (p.OW - rw + sow - 1) / sow;
  const int q0h = th * TH, q0w = tw * TW;
  if (qd >= Qd || q0h >= Qh || q0w >= Qw) return;

  int emin_d, emax_d, nvd, emin_h, emax_h, nvh, emin_w, emax_w, nvw;
  dim_range(p.kd, p.sd, p.pd, p.dd, rd, p.transposed, emin_d, emax_d, nvd);
  dim_range(p.kh, p.sh, p.ph, p.dh, rh, p.transposed, emin_h, emax_h, nvh);
  dim_range(p.kw, p.sw, p.pw, p.dw, rw, p.transposed, emin_w, emax_w, nvw);
  const int nvalid = nvd * nvh * nvw;

  const int od = qd * sod + rd;
  const long long out_plane = (long long)p.OH * p.OW;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;

  if (nvalid > 0) {
    const int ext_d = emax_d - emin_d + 1;
    const int ext_h = (TH - 1) * sxh + (emax_h - emin_h) + 1;
    const int ext_w = (TW - 1) * sxw + (emax_w - emin_w) + 1;
    const int planeStride = ext_h * ext_w;
    const int chanStride = ext_d * planeStride;
    const int i0d = qd * sxd + emin_d, i0h = q0h * sxh + emin_h, i0w = q0w * sxw + emin_w;

    if (tid == 0) {
      int nv = 0;
      for (int a = 0; a < p.kd; ++a) {
        int ed;
        if (!tap_e(a, p.kd, p.sd, p.pd, p.dd, rd, p.transposed, ed)) continue;
        for (int bb = 0; bb < p.kh; ++bb) {
          int eh;
          if (!tap_e(bb, p.kh, p.sh, p.ph, p.dh, rh, p.transposed, eh)) continue;
          for (int c = 0; c < p.kw; ++c) {
            int ew;
            if (!tap_e(c, p.kw, p.sw, p.pw, p.dw, rw, p.transposed, ew)) continue;
            s_tapoff[nv] = ((ed - emin_d) * ext_h + (eh - emin_h)) * ext_w + (ew - emin_w);
            s_tapw[nv] = (a * p.kh + bb) * p.kw + c;
            ++nv;
          }
        }
      }
      s_nv[0] = nv;
    }

    int lanebase[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) lanebase[t] = ((wave * NT + t) * sxh) * ext_w + l31 * sxw + hh * chanStride;

    const int in_rows = CC * ext_d * ext_h;
    const int rows_per_chan = ext_d * ext_h;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform row bookkeeping on the scalar unit
    const long long x_chan = (long long)p.ID * p.IH * p.IW;
    const float* xn = x + (long long)n * p.C * x_chan;
    for (int rowid = tid; rowid < in_rows; rowid += 256) {
      const int cc = rowid / rows_per_chan;
      const int rem = rowid - cc * rows_per_chan;
      const int pl = rem / ext_h;
      const int rr = rem - pl * ext_h;
      s_rowoff[rowid] = (int)((long long)cc * x_chan + ((long long)pl * p.IH + rr) * p.IW);
      s_rowpr[rowid] = (pl << 16) | rr | (cc << 24);
    }

    for (int c0 = 0; c0 < p.C; c0 += CC) {
      __syncthreads();   // previous chunk fully consumed (also orders the tap / row tables on the first trip)
      const float* xbase = xn + (long long)c0 * x_chan + ((long long)i0d * p.IH + i0h) * p.IW;
      // ---- stage the input patch: one LDS row per (channel, plane, row), lanes along W.  SU rows are fetched
      //      back-to-back before any is written so each wave keeps SU (x2) global loads in flight.
      constexpr int SU = 8;
      for (int r0 = wave_u * SU; r0 < in_rows; r0 += 4 * SU) {
        float v0[SU], v1[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int rowid = r0 + u;
          const int rsafe = rowid < in_rows ? rowid : 0;
          const int pr = s_rowpr[rsafe];
          const int cc = pr >> 24, pl = (pr >> 16) & 0xff, rr = pr & 0xffff;
          const int ic = c0 + cc, id = i0d + pl, ih = i0h + rr;
          const bool rowok = (rowid < in_rows) && (ic < p.C) && (id >= 0) && (id < p.ID) && (ih >= 0) && (ih < p.IH);
          const float* src = xbase + s_rowoff[rsafe];
          const int iw0 = i0w + lane, iw1 = iw0 + 64;
          v0[u] = (rowok && lane < ext_w && iw0 >= 0 && iw0 < p.IW) ? src[iw0] : 0.f;
          v1[u] = (rowok && lane + 64 < ext_w && iw1 >= 0 && iw1 < p.IW) ? src[iw1] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int rowid = r0 + u;
          if (rowid < in_rows) {
            float* dst = s_in + rowid * ext_w;   // rows are contiguous: cc*chanStride + rem*ext_w == rowid*ext_w
            if (lane < ext_w) dst[lane] = v0[u];
            if (lane + 64 < ext_w) dst[lane + 64] = v1[u];
          }
        }
      }
      const int nv = nvalid;
      __syncthreads();
      // ---- MFMA over (tap, channel pair); the operands of tap slot+1 are fetched from LDS while the MFMAs of
      //      tap `slot` issue (software pipeline, two register sets)
      // The weight fragments (A operand) come straight from the repacked [tap][channel][KT] tensor in global memory: every
      // workgroup reads the same <= 442 KB, so they are L2-resident, each half-wave load is one contiguous 128-B segment,
      // and keeping them out of LDS leaves room for more resident workgroups.
      float a_cur[CC / 2][MT], b_cur[CC / 2][NT], a_n1[CC / 2][MT], a_n2[CC / 2][MT], b_nxt[CC / 2][NT];
      auto load_a = [&](int slot, float (&dst)[CC / 2][MT]) {
        const int sl = slot < nv ? slot : nv - 1;
        const float* wrow = wt + ((long long)s_tapw[sl] * p.C + c0 + hh) * KT + l31;
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp) {
          const bool cok = c0 + 2 * cp + hh < p.C;
#pragma unroll
          for (int m = 0; m < MT; ++m) dst[cp][m] = cok ? wrow[(2 * cp) * KT + m * 32] : 0.f;
        }
      };
      load_a(0, a_cur);
      load_a(1, a_n1);
      {
        const int toff = s_tapoff[0];
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp)
#pragma unroll
          for (int t = 0; t < NT; ++t) b_cur[cp][t] = s_in[lanebase[t] + (2 * cp) * chanStride + toff];
      }
      for (int slot = 0; slot < nv; ++slot) {
        load_a(slot + 2, a_n2);                       // weights: two taps ahead (L2 latency)
        if (slot + 1 < nv) {                          // input patch: one tap ahead (LDS latency)
          const int toff = s_tapoff[slot + 1];
#pragma unroll
          for (int cp = 0; cp < CC / 2; ++cp)
#pragma unroll
            for (int t = 0; t < NT; ++t) b_nxt[cp][t] = s_in[lanebase[t] + (2 * cp) * chanStride + toff];
        }
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[cp][m], b_cur[cp][t], acc[m][t], 0, 0, 0);
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            a_cur[cp][m] = a_n1[cp][m];
            a_n1[cp][m] = a_n2[cp][m];
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) b_cur[cp][t] = b_nxt[cp][t];
        }
      }
    }
  }

  // ---- epilogue: D row = (j&3) + 8*(j>>2) + 4*(lane>>5), col = lane&31
  const int qw = q0w + l31;
  const int ow = qw * sow + rw;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int qh = q0h + wave * NT + t;
    if (qh >= Qh || qw >= Qw) continue;
    const int oh = qh * soh + rh;
    const long long pos = ((long long)od * p.OH + oh) * p.OW + ow;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (k < p.K) {
          float v = acc[m][t][j];
          if (bias) v += bias[p.k0 + k];
          out[((long long)n * p.Ktot + p.k0 + k) * p.OD * out_plane + pos] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  dW[k][c][t] += sum_{n,q} g[n,k,q] * x[n,c, q*s - p + t*dil]
// D[row = k][col = (c,t)],  reduction (MFMA K dim) over positions.
struct WgP {
  int N, C, K;
  int ID, IH, IW;   // x ("big grid") dims
  int QD, QH, QW;   // g ("small grid") dims
  int kd, kh, kw, T;
  int sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int Ktot, k0;     // g has Ktot channels; this launch covers [k0, k0 + K)
  int CCW;          // channels per block
  int es;           // polyphase factor: > 1 = dilation handled as es*es interleaved dilation-1 problems (dh = dw = 1 here)
  int nchunk;       // position chunks
  int tilesH, tilesW;
  long long ntiles;
};


// WTH rows of 32 positions per tile (8-row tiles for K <= 32 measured slower: 49 vs 52 TFLOP/s)
// WNT column tiles (of 32 (c,t) pairs) per wave (4 tiles for K <= 32 measured no faster: 48.5 vs 52 TFLOP/s)
template <int MT, int WTH = 4, int WNT = 2>
__global__ __launch_bounds__(256, (MT == 1 ? 4 : (MT == 2 ? 3 : 2))) void conv_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                         float* __restrict__ dw, WgP p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int KT = 32 * MT;
  const int ext_d = (p.kd - 1) * p.dd + 1;
  const int ext_h = (WTH - 1) * p.sh + (p.kh - 1) * p.dh + 1;
  const int ext_w = (TW - 1) * p.sw + (p.kw - 1) * p.dw + 1;
  const int planeStride = ext_h * ext_w;
  const int chanStride = ext_d * planeStride;
  float* s_x = smem;                         // [CCW][chanStride]
  constexpr int WPT = WTH * TW;
  float* s_g = s_x + p.CCW * chanStride;     // [KT][WPT+1]
  constexpr int GS = WPT + 1;
  int* s_rowoff = (int*)(s_g + KT * GS);     // per staged row: source offset relative to (n, c0, i0d, i0h)
  int* s_rowpr = s_rowoff + p.CCW * ext_d * ext_h;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int cchunk = blockIdx.x / p.nchunk;
  const int pchunk = blockIdx.x % p.nchunk;
  const int c0 = cchunk * p.CCW;
  const int ncc = min(p.CCW, p.C - c0);
  const int ncol = ncc * p.T;

  // per-lane column descriptors
  int colbase[WNT];
  bool colok[WNT];
#pragma unroll
  for (int t = 0; t < WNT; ++t) {
    const int coln = (wave * WNT + t) * 32 + l31;
    colok[t] = coln < ncol;
    const int cn = colok[t] ? coln : 0;
    const int cc = cn / p.T;
    const int tap = cn - cc * p.T;
    const int tw_ = tap % p.kw;
    const int th_ = (tap / p.kw) % p.kh;
    const int td_ = tap / (p.kw * p.kh);
    colbase[t] = cc * chanStride + (td_ * p.dd * ext_h + th_ * p.dh) * ext_w + tw_ * p.dw + hh * p.sw;
  }

  f32x16 acc[MT][WNT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < WNT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;

  const long long x_chan = (long long)p.ID * p.IH * p.IW;
  const long long g_chan = (long long)p.QD * p.QH * p.QW;
  const int rows_per_chan = ext_d * ext_h;
  for (int rowid = tid; rowid < ncc * rows_per_chan; rowid += 256) {
    const int cc = rowid / rows_per_chan;
    const int rem = rowid - cc * rows_per_chan;
    const int pl = rem / ext_h;
    const int rr = rem - pl * ext_h;
    s_rowoff[rowid] = (int)((long long)cc * x_chan + ((long long)pl * p.IH + (long long)rr * p.es) * p.IW);
    s_rowpr[rowid] = (pl << 16) | rr;
  }

  for (long long tile = pchunk; tile < p.ntiles; tile += p.nchunk) {
    long long b = tile;
    const int tw = (int)(b % p.tilesW); b /= p.tilesW;
    const int th = (int)(b % p.tilesH); b /= p.tilesH;
    int fh = 0, fw = 0;                        // phase of this tile (polyphase dilation: positions fh + es*i, fw + es*j)
    if (p.es > 1) {
      fw = (int)(b % p.es); b /= p.es;
      fh = (int)(b % p.es); b /= p.es;
    }
    const int qd = (int)(b % p.QD);
    const int n = (int)(b / p.QD);
    const int q0h = th * WTH, q0w = tw * TW;   // tile origin in phase coordinates
    const int es = p.es;
    const int i0d = qd * p.sd - p.pd, i0h = fh + es * q0h * p.sh - p.ph, i0w = fw + es * q0w * p.sw - p.pw;
    __syncthreads();
    // stage x patch (SU rows fetched back-to-back, then written)
    constexpr int SU = 8;
    const float* xbase = x + ((long long)n * p.C + c0) * x_chan + ((long long)i0d * p.IH + i0h) * p.IW;
    const int xrows = ncc * rows_per_chan;
    for (int r0 = wave_u * SU; r0 < xrows; r0 += 4 * SU) {
      float v0[SU], v1[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int rowid = r0 + u;
        const int rsafe = rowid < xrows ? rowid : 0;
        const int pr = s_rowpr[rsafe];
        const int pl = pr >> 16, rr = pr & 0xffff;
        const int id = i0d + pl, ih = i0h + es * rr;
        const bool rowok = (rowid < xrows) && (id >= 0) && (id < p.ID) && (ih >= 0) && (ih < p.IH);
        const float* src = xbase + s_rowoff[rsafe];
        const int iw0 = i0w + es * lane, iw1 = iw0 + es * 64;
        v0[u] = (rowok && lane < ext_w && iw0 >= 0 && iw0 < p.IW) ? src[iw0] : 0.f;
        v1[u] = (rowok && lane + 64 < ext_w && iw1 >= 0 && iw1 < p.IW) ? src[iw1] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int rowid = r0 + u;
        if (rowid < xrows) {
          float* dst = s_x + rowid * ext_w;
          if (lane < ext_w) dst[lane] = v0[u];
          if (lane + 64 < ext_w) dst[lane + 64] = v1[u];
        }
      }
    }
    // stage g tile: [k][row*32+col]; two (k, row) pairs per wave instruction (one per half)
    const float* gn = g + ((long long)n * p.Ktot + p.k0) * g_chan + (long long)qd * p.QH * p.QW;
    for (int r0 = wave_u * SU; r0 < KT * WTH / 2; r0 += 4 * SU) {
      float gv[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int rowid = 2 * (r0 + u) + hh;
        const int k = rowid / WTH;
        const int r = rowid - k * WTH;
        const int qh = fh + es * (q0h + r), qw = fw + es * (q0w + l31);
        gv[u] = (rowid < KT * WTH && k < p.K && qh < p.QH && qw < p.QW) ? gn[(long long)k * g_chan + (long long)qh * p.QW + qw] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int rowid = 2 * (r0 + u) + hh;
        if (rowid < KT * WTH) {
          const int k = rowid / WTH;
          const int r = rowid - k * WTH;
          s_g[k * GS + r * 32 + l31] = gv[u];
        }
      }
    }
    __syncthreads();
    for (int r = 0; r < WTH; ++r) {
      const int rowoff = (r * p.sh) * ext_w;
#pragma unroll 4
      for (int cs = 0; cs < 16; ++cs) {
        const int pos = r * 32 + 2 * cs + hh;
        float a[MT], bv[WNT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = s_g[(m * 32 + l31) * GS + pos];
#pragma unroll
        for (int t = 0; t < WNT; ++t) bv[t] = s_x[colbase[t] + rowoff + (2 * cs) * p.sw];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < WNT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[t], acc[m][t], 0, 0, 0);
      }
    }
  }
  // epilogue: atomics into dW[k][c][t]
#pragma unroll
  for (int t = 0; t < WNT; ++t) {
    if (!colok[t]) continue;
    const int coln = (wave * WNT + t) * 32 + l31;   // = cc*T + tap
    const long long base = (long long)c0 * p.T + coln;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (k < p.K) atomicAdd(dw + (long long)k * p.C * p.T + base, acc[m][t][j]);
      }
  }
}

int out_dim(int I, int k, int s, int p, int d) { return (I + 2 * p - (d * (k - 1) + 1)) / s + 1; }

template <int MT, int CC, int NT>
int launch_igemm(const float* x, const float* wt, const float* bias, float* out, const ConvP& p, size_t lds, hipStream_t st) {
  const long long blocks = (long long)p.ncls * p.N * p.QD * p.tilesH * p.tilesW;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DPF_ERR_INVALID_ARG;
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)conv_igemm_kernel<MT, CC, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return DPF_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<MT, CC, NT>), dim3((unsigned)blocks), dim3(256), lds, st, x, wt, bias, out, p);
  return dpf_check_launch();
}

int conv_launch(const float* x, const float* wt_ws, const float* bias, float* out, ConvP p, hipStream_t st);

int conv_common(const float* x, const float* w, const float* bias, float* out, float* wt_ws, ConvP p, int repack_mode,
                int wA, int wB, hipStream_t st, int Ktot = 0, int accumulate = 0) {
  const int T = p.kd * p.kh * p.kw;
  if (T > MAXT || T < 1) return DPF_ERR_UNSUPPORTED;
  // one launch covers up to 128 output channels (4 MFMA row tiles); wider outputs are split
  const int Kfull = p.K;
  p.Ktot = Ktot > 0 ? Ktot : Kfull;          // channels of the output tensor (>= the Kfull channels this call computes)
  for (int k0 = 0; k0 < Kfull; k0 += 128) {
    const int Kc = Kfull - k0 < 128 ? Kfull - k0 : 128;
    {   // LDS-DMA double-buffered kernel where the shape is eligible (conv_igemm2.hip)
      DpfConvDesc d{p.N, p.C, Kc, p.Ktot, k0, p.ID, p.IH, p.IW, p.OD, p.OH, p.OW, p.kd, p.kh, p.kw, p.sd, p.sh, p.sw,
                    p.pd, p.ph, p.pw, p.dd, p.dh, p.dw, p.transposed, wA, wB, repack_mode, accumulate};
      if (T == 1 && !accumulate) {                   // pointwise: HBM-bound direct kernel (conv_pointwise.hip)
        const int rcp = dpf_pointwise_conv(x, w, bias, out, d, st);
        if (rcp == DPF_OK) continue;
        if (rcp != DPF_ERR_UNSUPPORTED) return rcp;
      }
      const int rc2 = dpf_igemm2_conv(x, w, bias, out, wt_ws, d, st);
      if (rc2 == DPF_OK) continue;
      if (rc2 != DPF_ERR_UNSUPPORTED) return rc2;
    }
    if (accumulate) return DPF_ERR_UNSUPPORTED;      // the first-generation kernel only stores
    const int KT = 32 * ((Kc + 31) / 32);
    const long long total = (long long)T * p.C * KT;
    hipLaunchKernelGGL(repack_weights_kernel, dim3(dpf_ew_grid(total)), dim3(256), 0, st, w, wt_ws, wA, wB, T, KT, repack_mode, k0, Kc);
    if (dpf_check_launch() != DPF_OK) return DPF_ERR_LAUNCH;
    p.K = Kc;
    p.k0 = k0;
    const int rc = conv_launch(x, wt_ws, bias, out, p, st);
    if (rc != DPF_OK) return rc;
  }
  return DPF_OK;
}

int conv_launch(const float* x, const float* wt_ws, const float* bias, float* out, ConvP p, hipStream_t st) {
  const int T = p.kd * p.kh * p.kw;
  const int MT = (p.K + 31) / 32;
  const int KT = 32 * MT;
  const int NT = MT == 1 ? 4 : 2;   // narrow outputs: more position tiles per wave so each weight fragment feeds 4 MFMAs
  const int TH = 4 * NT;
  // tile geometry (host worst case over classes)
  int ext_d, ext_h, ext_w, ntmax;
  if (!p.transposed) {
    p.ncls = 1;
    p.QD = p.OD;
    p.tilesH = dpf_div_up(p.OH, TH);
    p.tilesW = dpf_div_up(p.OW, TW);
    ext_d = (p.kd - 1) * p.dd + 1;
    ext_h = (TH - 1) * p.sh + (p.kh - 1) * p.dh + 1;
    ext_w = (TW - 1) * p.sw + (p.kw - 1) * p.dw + 1;
    ntmax = T;
  } else {
    p.ncls = p.sd * p.sh * p.sw;
    p.QD = dpf_div_up(p.OD, p.sd);
    p.tilesH = dpf_div_up(dpf_div_up(p.OH, p.sh), TH);
    p.tilesW = dpf_div_up(dpf_div_up(p.OW, p.sw), TW);
    ext_d = ((p.kd - 1) * p.dd) / p.sd + 1;
    ext_h = (TH - 1) + ((p.kh - 1) * p.dh) / p.sh + 1;
    ext_w = (TW - 1) + ((p.kw - 1) * p.dw) / p.sw + 1;
    auto nvmax = [](int k, int s, int d) {   // max taps of one parity class along a dim
      int best = 0;
      for (int r = 0; r < s; ++r) {
        int c = 0;
        for (int t = 0; t < k; ++t) c += ((((r - t * d) % s) + s) % s == 0);
        if (c > best) best = c;
      }
      // pad shifts the classes but not their sizes
      return best;
    };
    ntmax = nvmax(p.kd, p.sd, p.dd) * nvmax(p.kh, p.sh, p.dh) * nvmax(p.kw, p.sw, p.dw);
  }
  if (ext_w > 128) return DPF_ERR_UNSUPPORTED;   // staging handles two 64-lane column groups
  p.chanStrideMax = ext_d * ext_h * ext_w;
  p.ntmax = ntmax;
  auto lds_bytes = [&](int CC) { return (size_t)(CC * p.chanStrideMax + 2 * MAXT + 4 + 2 * CC * ext_d * ext_h) * sizeof(float); };
  int CC = 8;
  constexpr int lds_cap = 20000;
  if (lds_bytes(8) > (size_t)lds_cap || p.C <= 4) CC = 4;
  const size_t lds = lds_bytes(CC);
  if (lds > 160 * 1024) return DPF_ERR_UNSUPPORTED;
  p.maxrows = CC * ext_d * ext_h;
#define DPF_IG(M, Cc, Nt) return launch_igemm<M, Cc, Nt>(x, wt_ws, bias, out, p, lds, st)
  if (CC == 8) {
    switch (MT) { case 1: DPF_IG(1, 8, 4); case 2: DPF_IG(2, 8, 2); case 3: DPF_IG(3, 8, 2); default: DPF_IG(4, 8, 2); }
  } else {
    switch (MT) { case 1: DPF_IG(1, 4, 4); case 2: DPF_IG(2, 4, 2); case 3: DPF_IG(3, 4, 2); default: DPF_IG(4, 4, 2); }
  }
#undef DPF_IG
}

}  // namespace

int dpf_conv_wgrad_slice(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int kbeg, int kcount, int QD,
                         int QH, int QW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_,
                         void* stream);

namespace { int g_operand_bf16 = 0; }
int dpf_conv_operand_bf16() { return g_operand_bf16; }
namespace { int g_f32_x9 = -1; }
int dpf_conv_f32_x9() {
  if (g_f32_x9 < 0) {
    const int v = getenv("DPF_F32_X9") ? atoi(getenv("DPF_F32_X9")) : 2;
    g_f32_x9 = v < 0 ? 0 : (v > 2 ? 2 : v);
  }
  return g_f32_x9;
}
namespace { int g_h3_guard = 1; }
int dpf_h3_range_guard() { return g_h3_guard; }

extern "C" {

// 0: exact fp32 operands (default); 1: the dense convolution kernels (forward, stride-1 data gradient, weight gradient) round their
// operands to bf16 (RNE) while staging them, accumulate and store in fp32.  Process-wide; the host side sets it around each launch.
int dpf_set_conv_operand_precision(int bf16) {
  g_operand_bf16 = bf16 ? 1 : 0;
  return DPF_OK;
}
int dpf_get_conv_operand_precision(void) { return g_operand_bf16; }
int dpf_set_f32_matrix_path(int split_bf16) {
  g_f32_x9 = split_bf16 < 0 ? 0 : (split_bf16 > 2 ? 2 : split_bf16);
  return DPF_OK;
}
int dpf_get_f32_matrix_path(void) { return dpf_conv_f32_x9(); }
// diagnostic: 0 switches the position guard of the f16-component convolutions off (round 5's behaviour) so that a test can show what it buys
int dpf_debug_set_range_guard(int on) {
  g_h3_guard = on ? 1 : 0;
  return DPF_OK;
}

// workspace (floats) needed for the repacked weights of a conv with `T` taps, `reduce` reduction channels
// and `outc` output channels
long long dpf_conv_workspace_floats(int T, int reduce, int outc) {
  const long long a = (long long)T * reduce * (((outc + 31) / 32) * 32), b = dpf_igemm2_workspace_floats(T, reduce, outc);
  return a > b ? a : b;
}

// x [N,C,ID,IH,IW], w [K,C,kd,kh,kw], bias [K] or NULL, out [N,K,OD,OH,OW]
int dpf_conv_forward(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                     int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
                     void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !w || !out || !ws || N <= 0 || C <= 0 || K <= 0) return DPF_ERR_INVALID_ARG;
  ConvP p{};
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.OD = out_dim(ID, kd, sd, pd, dd); p.OH = out_dim(IH, kh, sh, ph, dh); p.OW = out_dim(IW, kw, sw, pw, dw);
  if (p.OD <= 0 || p.OH <= 0 || p.OW <= 0) return DPF_ERR_INVALID_ARG;
  p.transposed = 0;
  return conv_common(x, w, bias, out, ws, p, /*mode*/ 0, K, C, (hipStream_t)stream);
}

// dpf_conv_forward that also leaves, per position tile, the (sum, sum of squares) of every output channel in `slab`
// ([*parts_host][K][2] doubles, capacity dpf_conv_stats_slab_doubles) for the BatchNorm that follows (dpf_bn_finalize_partials): the
// separate statistics pass over the output tensor disappears.  DPF_ERR_UNSUPPORTED when the shape does not run on the LDS-DMA
// kernel (K > 128, rows not 16-byte aligned, 1x1 kernels ...): the caller then uses dpf_conv_forward + dpf_bn_stats.
long long dpf_conv_stats_slab_doubles(int N, int K, int OD, int OH, int OW) {
  return 2LL * K * ((long long)N * OD * dpf_div_up(OH, 8) * dpf_div_up(OW, 32) + 64);     // + the 64 folded rows of the finalize step
}

int dpf_conv_forward_stats(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                           int K, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
                           double* slab, long long slab_doubles, int* parts_host, void* stream) {
  dpf_clear_error();
  if (!x || !w || !out || !ws || !slab || !parts_host || N <= 0 || C <= 0 || K <= 0) return DPF_ERR_INVALID_ARG;
  const int OD = out_dim(ID, kd, sd, pd, dd), OH = out_dim(IH, kh, sh, ph, dh), OW = out_dim(IW, kw, sw, pw, dw);
  if (OD <= 0 || OH <= 0 || OW <= 0) return DPF_ERR_INVALID_ARG;
  DpfConvDesc d{N, C, K, K, 0, ID, IH, IW, OD, OH, OW, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw, 0, K, C, 0};
  DpfConvStats stats{slab, slab_doubles - 2LL * K * 64, 0};
  const int rc = dpf_igemm2_conv(x, w, bias, out, ws, d, (hipStream_t)stream, &stats);
  if (rc == DPF_OK) *parts_host = stats.parts;
  return rc;
}

// Transposed convolution.  x [N,C,ID,IH,IW] lives on the strided (small) grid, out [N,K,OD,OH,OW] on the dense grid;
// (OD,OH,OW) are given by the caller (output_padding ambiguity).  `w_is_conv_layout` = 1: w is a forward-conv weight
// [C(x chans = conv out), K(out chans = conv in), T] and this call is that conv's data gradient;
// = 0: w is an nn.ConvTranspose3d weight [C_in = C, C_out = K, T].  Both are [C][K][T] in memory.
int dpf_conv_transpose(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                       int K, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                       int dd, int dh, int dw, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!x || !w || !out || !ws || N <= 0 || C <= 0 || K <= 0) return DPF_ERR_INVALID_ARG;
  ConvP p{};
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW; p.OD = OD; p.OH = OH; p.OW = OW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.transposed = 1;
  // w[C][K][T]: reduce = A (=C), out = B (=K)
  return conv_common(x, w, bias, out, ws, p, /*mode*/ 1, C, K, (hipStream_t)stream);
}

// As dpf_conv_transpose for an output tensor (and weight) of Ktot channels of which only the first K are computed (the others
// are left untouched): data gradients whose trailing input channels have no consumer (the constant XYZ channels of the ANM volume).
int dpf_conv_transpose_ex(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                          int K, int Ktot, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                          int dd, int dh, int dw, void* stream) {
  dpf_clear_error();
  if (!x || !w || !out || !ws || N <= 0 || C <= 0 || K <= 0 || Ktot < K) return DPF_ERR_INVALID_ARG;
  ConvP p{};
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW; p.OD = OD; p.OH = OH; p.OW = OW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.transposed = 1;
  return conv_common(x, w, bias, out, ws, p, /*mode*/ 1, C, Ktot, (hipStream_t)stream, Ktot);
}

// dpf_conv_transpose_ex with out += result when accumulate != 0: the data gradients of several convolutions that read the same tensor
// (the three dilated branches of a DPBlock, modules.py:43-45) are summed in the kernel epilogue instead of by separate add passes.
// DPF_ERR_UNSUPPORTED (nothing written) when the shape would not run on the LDS-DMA kernel: compute into a temporary and add.
int dpf_conv_transpose_acc(const float* x, const float* w, const float* bias, float* out, float* ws, int N, int C, int ID, int IH, int IW,
                           int K, int Ktot, int OD, int OH, int OW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                           int dd, int dh, int dw, int accumulate, void* stream) {
  dpf_clear_error();
  if (!x || !w || !out || !ws || N <= 0 || C <= 0 || K <= 0 || Ktot < K) return DPF_ERR_INVALID_ARG;
  if (accumulate && K > 128) return DPF_ERR_UNSUPPORTED;      // split launches: a failure midway would leave a partial sum
  ConvP p{};
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW; p.OD = OD; p.OH = OH; p.OW = OW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw;
  p.transposed = 1;
  return conv_common(x, w, bias, out, ws, p, /*mode*/ 1, C, Ktot, (hipStream_t)stream, Ktot, accumulate ? 1 : 0);
}

// Forward conv with the weight stored transposed-conv style w[K_reduce=C? ...]:
// data gradient of nn.ConvTranspose3d: out[n,ci,q] = sum_{co,t} w[ci][co][t] * g[n,co, q*s - p + t*dil]
// i.e. a forward conv whose weight is indexed [out][reduce][t] -- identical to dpf_conv_forward (w[K][C][T]).

// dW[K][C][T] += ...   (dw must be zero-initialised or hold the running gradient)
// g [N,K,QD,QH,QW] on the small grid, x [N,C,ID,IH,IW] on the dense grid.
long long dpf_conv_wgrad_workspace_floats(int T, int C, int K) {
  const long long a = dpf_wgrad2_workspace_floats(T, C, K), b = T == 1 ? dpf_pointwise_wgrad_workspace_floats(C, K < 128 ? K : 128) : 0;
  return a > b ? a : b;
}

// as dpf_conv_wgrad below, with caller scratch (dpf_conv_wgrad_workspace_floats floats): eligible shapes run the LDS-DMA kernel
// with a deterministic slab reduction (conv_wgrad2.hip) instead of float atomics
int dpf_conv_wgrad_ws(const float* g, const float* x, float* dw, float* ws, long long ws_floats, int N, int C, int ID, int IH, int IW, int K,
                      int QD, int QH, int QW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_,
                      int accumulate, void* stream) {
  dpf_clear_error();
  if (!g || !x || !dw || N <= 0 || C <= 0 || K <= 0) return DPF_ERR_INVALID_ARG;
  const int T = kd * kh * kw;
  for (int k0 = 0; k0 < K; k0 += 128) {
    const int Kc = K - k0 < 128 ? K - k0 : 128;
    float* dwk = dw + (long long)k0 * C * T;
    int rc = DPF_ERR_UNSUPPORTED;
    if (ws) {
      DpfWgradDesc d{N, C, Kc, K, k0, ID, IH, IW, QD, QH, QW, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw_};
      if (T == 1) rc = dpf_pointwise_wgrad(g, x, dwk, ws, ws_floats, d, accumulate, (hipStream_t)stream);
      if (rc == DPF_ERR_UNSUPPORTED) rc = dpf_wgrad2(g, x, dwk, ws, ws_floats, d, accumulate, (hipStream_t)stream);
    }
    if (rc == DPF_ERR_UNSUPPORTED) {
      if (!accumulate && hipMemsetAsync(dwk, 0, sizeof(float) * (size_t)Kc * C * T, (hipStream_t)stream) != hipSuccess) return DPF_ERR_LAUNCH;
      // generic kernel on this channel slice: g viewed with Ktot = K channels, slice [k0, k0 + Kc)
      rc = dpf_conv_wgrad_slice(g, x, dwk, N, C, ID, IH, IW, K, k0, Kc, QD, QH, QW, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw_, stream);
    }
    if (rc != DPF_OK) return rc;
  }
  return DPF_OK;
}

int dpf_conv_wgrad(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int QD, int QH, int QW,
                   int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_, void* stream) {
  dpf_clear_error();
  return dpf_conv_wgrad_slice(g, x, dw, N, C, ID, IH, IW, K, 0, K, QD, QH, QW, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw_, stream);
}

}  // extern "C"

// generic (first-generation) weight gradient of the g-channel slice [kbeg, kbeg + kcount) of a g tensor with K channels; dw points at
// row kbeg of dW
int dpf_conv_wgrad_slice(const float* g, const float* x, float* dw, int N, int C, int ID, int IH, int IW, int K, int kbeg, int kcount, int QD,
                         int QH, int QW, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw_,
                         void* stream) {   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!g || !x || !dw || N <= 0 || C <= 0 || K <= 0) return DPF_ERR_INVALID_ARG;
  WgP p{};
  p.N = N; p.C = C; p.K = K; p.ID = ID; p.IH = IH; p.IW = IW; p.QD = QD; p.QH = QH; p.QW = QW;
  p.kd = kd; p.kh = kh; p.kw = kw; p.T = kd * kh * kw;
  p.sd = sd; p.sh = sh; p.sw = sw; p.pd = pd; p.ph = ph; p.pw = pw; p.dd = dd; p.dh = dh; p.dw = dw_;
  if (p.T > MAXT) return DPF_ERR_UNSUPPORTED;
  p.Ktot = K;
  const int Kend = kbeg + kcount;
  // Dilated stride-1 convolutions: output (y, x) only meets inputs of its own residue class mod d, so the problem splits into
  // d*d interleaved dilation-1 problems.  Tiles then carry a (kh-1)-wide halo instead of (kh-1)*d, at the price of strided
  // (every d-th element) staging loads, which L2 absorbs.
  p.es = 1;
  if (sh == 1 && sw == 1 && dh == dw_ && dh > 1) {
    p.es = dh;
    p.dh = p.dw = 1;
    dh = dw_ = 1;
  }
  const int QHp = dpf_div_up(QH, p.es), QWp = dpf_div_up(QW, p.es);   // extent of one phase
  for (int k0 = kbeg; k0 < Kend; k0 += 128) {   // one launch covers up to 128 g-channels (4 MFMA row tiles)
  p.k0 = k0;
  p.K = Kend - k0 < 128 ? Kend - k0 : 128;
  float* dwk = dw + (long long)(k0 - kbeg) * C * p.T;
  const int MT = (p.K + 31) / 32;
  const int KT = 32 * MT;
  const int WNT = 2;
  int CCW = (4 * WNT * 32) / p.T;
  if (CCW > C) CCW = C;
  if (CCW < 1) CCW = 1;
  const int ext_d = (kd - 1) * dd + 1;
  const int WTH = 4, WPT = WTH * TW;
  const int ext_h = (WTH - 1) * sh + (kh - 1) * dh + 1;
  const int ext_w = (TW - 1) * sw + (kw - 1) * dw_ + 1;
  if (ext_w > 128) return DPF_ERR_UNSUPPORTED;
  auto lds_bytes = [&](int ccw) { return (size_t)(ccw * ext_d * ext_h * ext_w + KT * (WPT + 1) + 2 * ccw * ext_d * ext_h) * sizeof(float); };
  while (CCW > 1 && lds_bytes(CCW) > 96 * 1024) --CCW;
  if (lds_bytes(CCW) > 160 * 1024) return DPF_ERR_UNSUPPORTED;
  p.CCW = CCW;
  p.tilesH = dpf_div_up(QHp, WTH);
  p.tilesW = dpf_div_up(QWp, TW);
  p.ntiles = (long long)N * QD * p.es * p.es * p.tilesH * p.tilesW;
  const int cchunks = dpf_div_up(C, CCW);
  long long nchunk = 2048 / cchunks;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > p.ntiles) nchunk = p.ntiles;
  if (dpf_deterministic()) nchunk = 1;     // every dW address then receives ONE atomic add per launch: order-independent (dpf_common.h)
  p.nchunk = (int)nchunk;
  const size_t lds = lds_bytes(CCW);
  const dim3 grid((unsigned)(cchunks * p.nchunk));
  hipStream_t st = (hipStream_t)stream;
#define DPF_WG(M)                                                                                                           \
  {                                                                                                                         \
    if (lds > 48 * 1024 &&                                                                                                  \
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return DPF_ERR_LAUNCH;                                                                                                \
    hipLaunchKernelGGL((conv_wgrad_kernel<M>), grid, dim3(256), lds, st, g, x, dwk, p);                                      \
  }
  switch (MT) { case 1: DPF_WG(1); break; case 2: DPF_WG(2); break; case 3: DPF_WG(3); break; default: DPF_WG(4); break; }
#undef DPF_WG
  }
  return dpf_check_launch();
}
