// Weight gradient of the dense convolutions, second generation: LDS-DMA double buffering, position-split waves,
// deterministic slab reduction.
//
//   dW[k][c][t] += sum_{n,q} g[n,k,q] * x[n,c, q*s - p + t*dil]          (replaces cuDNN's wgrad behind nn.Conv2d / nn.Conv3d;
//   reference call sites as in conv_igemm.hip)
//
// D[row = k][col = (c, t)], reduction over positions on v_mfma_f32_32x32x2_f32 (exact f32).  A 256-thread workgroup owns a
// 32-row slice of K and NCT column tiles (a whole number of input channels x all taps, <= 7 tiles) and walks position tiles
// of 4 rows x 32.  Inside the workgroup the four waves split the POSITIONS (wave w = tile row w), not the columns: every wave
// accumulates all NCT tiles over its own row, so the matrix pipes are evenly loaded for any column count (the first-generation
// kernel and the column-split variant of this one idled 16 % of their MFMAs on ragged column chunks), and the four partial
// results are summed once, through LDS, after the last tile.
//   * both operands of a tile arrive by `global_load_lds_dwordx4` into one of two LDS buffers while the MFMAs of the previous
//     tile run; one vmcnt(0) + barrier per tile;
//   * the g tile keeps its global [k][position] order but its 16-byte slots are XOR-swizzled with (k & 15) through the DMA's
//     per-lane SOURCE address: a lane (= row k) fetches 4 consecutive positions with ONE conflict-free ds_read_b128 that feeds
//     4 x NCT MFMAs (the pairing of reduction indices inside an MFMA is free: lane half h takes positions 8j+4h .. 8j+4h+3);
//   * the x patch is the forward kernel's aligned 16-byte-segment image; plane / channel strides are padded (host search) so
//     the per-lane tap gather of the B operand spreads over the LDS banks;
//   * a workgroup's tiles are a CONTIGUOUS range ordered depth-fastest, so the planes / halo rows it re-reads and the g tiles
//     shared by the column chunks of the same positions are L2 hits (the strided assignment measured 4.8x the algorithmic bytes);
//   * no float atomics: every workgroup stores its partial dW to a slab, a second kernel adds the slabs in a fixed order ->
//     bitwise reproducible gradients.
#include "conv_internal.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int NLX = 8;      // x-patch DMA instructions per thread and tile (upper bound: 32 KB)
constexpr int WTH = 4;      // tile rows = waves
constexpr int SPR = WTH * 8;   // 16-byte slots per g row
constexpr int GFLOATS = 32 * SPR * 4;

__device__ float4 dpf_wg2_zero_page[4];   // zero-initialised at module load: source of out-of-bounds segments

struct W2P {
  int N, C, K, Ktot, k0;
  int ID, IH, IW, QD, QH, QW;
  int sd, sh, sw, pd, ph, pw;
  int kd, kh, kw, dd, dh, dw, T;
  int rstep, dhl;              // row step of a tile (dilated rows, below) and the tap-row distance inside the LDS image (dh / rstep)
  int ext_d, ext_h, RS, SR, PS, CS, PSseg, CSseg, colshift;   // x image: row / plane / channel strides (floats, segments)
  int CCW, cchunks, kslices, groups, nxseg;
  int tilesH, tilesW, nchunk;
  long long ntiles, per;       // tiles per position chunk (contiguous range)
  unsigned mCS, mPS, mSR;
  long long slab_stride;       // floats between position-chunk slabs  (K * C * T)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// four floats -> four bf16 (round to nearest even): two v_cvt_pk_bf16_f32
__device__ __forceinline__ s16x4 pack4_bf16(float a, float b, float c, float d) {
  const f32x2 lo = {a, b}, hi = {c, d};
  const u32x2 u = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2))};
  return __builtin_bit_cast(s16x4, u);
}

__device__ __forceinline__ void glds16(const float* gsrc, float* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// exact three-way split of 8 floats into packed bf16 (conv_internal.h: round-to-nearest split): v[i] = hi[i] + mid[i] + lo[i]
__device__ __forceinline__ void split8_bf16(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned a, b, c;
    dpf_split_pair(v[2 * q], v[2 * q + 1], a, b, c);
    h[q] = a; m[q] = b; l[q] = c;
  }
  hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

constexpr int w2_occ_of(int nct) {
  const int est = nct * 17 + 2 * NLX + 64;
  return est <= 128 ? 4 : (est <= 168 ? 3 : 2);
}
template <int NCT>
constexpr int w2_occ() { return w2_occ_of(NCT); }

// BF = true (operand precision "bf16"): the 4 positions a lane half holds of a group are exactly the 4 reduction indices a lane half
// feeds to v_mfma_f32_32x32x8_bf16, so a (group, column tile) unit is ONE bf16 MFMA on the RNE-rounded g / x values instead of four
// exact-f32 ones; staging, fp32 accumulation, slab fold are unchanged.
// SW1 = true: column stride 1 known at compile time -- the B-fragment reads s_x[colbase[t] + (8 j + i) * sw] become base + immediate (the
// 109 v_add_u32 per tile pair that computed those addresses sat between the MFMAs of the loop, and on this chip vector-ALU work and fp32
// MFMAs never overlap: tools/lean_probe2.hip)
// X9 = true (the default for operand precision "f32", DPF_F32_X9=0 turns it off): fp32 products on the bf16 matrix pipe.  Both operands
// are split EXACTLY into three bf16 terms (x = hi + mid + lo: three 8-bit truncations of the 24-bit significand), and a (16 positions x
// column tile) unit is the 9 partial products hi/mid/lo x hi/mid/lo as v_mfma_f32_32x32x16_bf16 -- every partial product is exact, the sum
// runs in the MFMA's fp32 accumulator, smallest terms first.  9 x 34 = 306 matrix-pipe clocks per 16 reduction indices instead of 8 x 64 =
// 512 on v_mfma_f32_32x32x2_f32, and, unlike the fp32 MFMA, the bf16 MFMA leaves the vector ALU free: the ~11 split instructions per value
// pair run in its shadow (tools/lean_probe2.hip: 4 v_fma_f32 per bf16 MFMA cost nothing, behind an fp32 MFMA they cost their full time).
// first of the nine partial products (smallest first) that is issued: conv_internal.h (default 3: six products)
// X9 = 2 (dpf_set_f32_matrix_path(2)): fp32 products from two f16 components per operand (conv_internal.h), three MFMAs per unit on
// v_mfma_f32_32x32x16_f16.  A component pair fits the 4 bytes of the fp32 value it came from, so a tile is converted IN PLACE, once per
// element instead of once per use (an x value is read by up to 27 taps): after the tile has landed every lane reads back the 16-byte
// segments it fetched itself, the workgroup agrees on the largest exponent of the x patch and of the g tile (the exchange rides on the
// tile barrier), and each lane rewrites its segments as (hi << 16 | lo) words of the values scaled by the running maxima; the
// accumulators are rescaled (exactly) when a maximum grows.  The unit loop then assembles an MFMA operand from 8 words with 8 v_perm_b32
// -- no conversion, no subtraction -- and the fragment addressing is the fp32 kernel's.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int W2_X9_FIRST = DPF_X9_FIRST;
template <int NCT, bool BF = false, bool SW1 = false, int X9 = 0>
__global__ __launch_bounds__(256, (X9 ? (NCT <= 2 ? 3 : 2) : w2_occ<NCT>())) void wgrad2_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                     float* __restrict__ slab, W2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups that walk the SAME tiles (the column chunks / k slices of one position chunk) share the g tiles and x patches:
  // they get the same blockIdx % 8 label, i.e. (observed round-robin dispatch) the same XCD and L2
  const int jb = blockIdx.x >> 3;
  const int pchunk = (jb / p.groups) * 8 + (blockIdx.x & 7);
  if (pchunk >= p.nchunk) return;
  const int gidx = jb % p.groups;
  const int cchunk = gidx % p.cchunks;
  const int kslice = gidx / p.cchunks;
  const int c0 = cchunk * p.CCW;
  const int ncc = min(p.CCW, p.C - c0);
  const int ncol = ncc * p.T;
  const int kbase = kslice * 32;                 // first g channel of this workgroup (relative to k0)
  const int krows = min(32, p.K - kbase);
  const int xFloats = p.CCW * p.CS;
  const int bufFloats = xFloats + GFLOATS;
  const float* zero = reinterpret_cast<const float*>(dpf_wg2_zero_page);

  // ---- column descriptors (B operand): lane -> (channel, tap) -> LDS offset of that tap for position 4*hh of this wave's row
  int colbase[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) {
    const int coln = t * 32 + l31;
    const int cn = coln < ncol ? coln : 0;
    const int cc = cn / p.T;
    const int tap = cn - cc * p.T;
    const int tw_ = tap % p.kw, th_ = (tap / p.kw) % p.kh, td_ = tap / (p.kw * p.kh);
    colbase[t] = cc * p.CS + td_ * p.dd * p.PS + (th_ * p.dhl + wave * p.sh) * p.RS + tw_ * p.dw + p.colshift + 4 * hh * p.sw;
  }

  // ---- x-patch DMA descriptors: flat segment f -> (cc, plane, row, seg); tile independent
  const long long x_chan = (long long)p.ID * p.IH * p.IW;
  const long long g_chan = (long long)p.QD * p.QH * p.QW;
  int xoff[NLX], xmeta[NLX];      // source offset relative to the tile origin; (channel << 24) | (plane << 16) | (row << 8) | seg, or -1 (no DMA)
#pragma unroll
  for (int j = 0; j < NLX; ++j) {
    const unsigned f = tid + 256 * j;
    const unsigned cc = (f * p.mCS) >> 20;
    const unsigned r1 = f - cc * p.CSseg;
    const unsigned pl = (r1 * p.mPS) >> 20;
    const unsigned r2 = r1 - pl * p.PSseg;
    const unsigned rr = (r2 * p.mSR) >> 20;
    const unsigned seg = r2 - rr * p.SR;
    const bool ok = (int)f < p.nxseg && (int)cc < ncc && (int)pl < p.ext_d && (int)rr < p.ext_h;
    xmeta[j] = ok ? (int)((cc << 24) | (pl << 16) | (rr << 8) | seg) : -1;     // (cc < 32: at most 7 x 32 / 9 channels per workgroup)
    xoff[j] = (int)((long long)cc * x_chan + ((long long)pl * p.IH + rr * p.rstep) * p.IW + 4 * seg);
  }
  // ---- g-tile DMA descriptors: physical slot P = k*SPR + sp (P = tid + 256 j) holds the logical slot sp ^ (k & 15) of row k;
  //      k = tid/SPR + 8 j, so the permutation depends on j only through its parity (two variants), the row advances by a scalar
  constexpr int KSTEP = 256 / SPR, NG = 32 / KSTEP;
  const int gk0 = tid / SPR;
  int glg[2], goff[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int lg = (tid % SPR) ^ ((gk0 + v * KSTEP) & 15);
    glg[v] = lg;
    goff[v] = (int)((long long)gk0 * g_chan + (long long)(lg >> 3) * p.rstep * p.QW + 4 * (lg & 7));
  }

  // tile cursor (depth fastest): decoded once, then advanced by one per issued tile -- no integer division in the tile loop
  int cur_qd, cur_tw, cur_th, cur_n;
  {
    long long r = (long long)pchunk * p.per;
    cur_qd = (int)(r % p.QD); r /= p.QD;
    cur_tw = (int)(r % p.tilesW); r /= p.tilesW;
    cur_th = (int)(r % p.tilesH);
    cur_n = (int)(r / p.tilesH);
  }

  auto issue = [&](int buf) {
    // rstep > 1 (rows dilated by dh, stride 1): tile rows are q0h + w * dh -- the rows of one dilation phase -- so the patch is kh + 3
    // image rows fetched dh apart instead of 3 + (kh - 1) * dh + 1 consecutive ones
    const int n = cur_n, qd = cur_qd, q0w = cur_tw * 32;
    const int q0h = p.rstep == 1 ? cur_th * WTH : (cur_th / p.rstep) * (WTH * p.rstep) + cur_th % p.rstep;
    if (++cur_qd == p.QD) {
      cur_qd = 0;
      if (++cur_tw == p.tilesW) {
        cur_tw = 0;
        if (++cur_th == p.tilesH) { cur_th = 0; ++cur_n; }
      }
    }
    float* dbase = smem + buf * bufFloats;
    const int i0d = qd * p.sd - p.pd, i0h = q0h * p.sh - p.ph, a0 = q0w * p.sw - p.pw - p.colshift;
    const float* xt = x + ((long long)n * p.C + c0) * x_chan + ((long long)i0d * p.IH + i0h) * p.IW + a0;
    const long long xzero = zero - xt;                           // offset of the zero page from this tile's origin (both 4-byte aligned)
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      if (j * 256 < p.nxseg) {                                   // wave-uniform
        const int m = xmeta[j];
        if (m >= 0) {
          // branch-free bounds test (unsigned compares, bitwise and): the short-circuit form compiles to nested exec-mask regions
          const unsigned id = (unsigned)(i0d + ((m >> 16) & 0xff)), ih = (unsigned)(i0h + ((m >> 8) & 0xff) * p.rstep), iw = (unsigned)(a0 + 4 * (m & 0xff));
          const bool ok = (id < (unsigned)p.ID) & (ih < (unsigned)p.IH) & (iw < (unsigned)p.IW);
          const long long so = ok ? (long long)xoff[j] : xzero;
          glds16(xt + so, dbase + (j * 256 + wave * 64) * 4);
        }
      }
    }
    const float* gt = g + ((long long)n * p.Ktot + p.k0 + kbase) * g_chan + ((long long)qd * p.QH + q0h) * p.QW + q0w;
    float* gbase = dbase + xFloats;
    bool gok[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) gok[v] = q0h + (glg[v] >> 3) * p.rstep < p.QH && q0w + 4 * (glg[v] & 7) < p.QW;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const bool ok = (gk0 + j * KSTEP < krows) & gok[j & 1];
      glds16(ok ? gt + (long long)j * KSTEP * g_chan + goff[j & 1] : zero, gbase + (j * 256 + wave * 64) * 4);
    }
  };

  f32x16 acc[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  // ---- X9 = 2: exponent scan and in-place conversion of the segments this lane fetched (same slots as issue()).
  // RANGE GUARD (conv_internal.h): an output element dW[k][c][t] sums over POSITIONS, so rows (g channels) and columns (x channels) of the
  // MFMA tile may be scaled independently -- every g row and every x channel of the workgroup carries its own running exponent
  // (s_ge[32], s_xe[32] in LDS; a lane keeps the exponents of the segments it fetches, of its columns and of its accumulator rows as
  // packed bytes).  The scan compares a segment's exponent with the cached one of its channel; only a lane that finds a larger one
  // touches LDS (ds_max + a flag), and only then -- the first tiles of a workgroup, then hardly ever -- the workgroup refreshes its
  // caches and rescales the accumulators (exactly) by row and column.  A weight-gradient element is then exact to fp32 relative to the
  // magnitudes of ITS OWN g channel and x channel, whatever the other channels of the tile hold.
  int* s_xe = reinterpret_cast<int*>(smem + 2 * bufFloats);        // running exponent of x channel cc (of this workgroup's CCW channels)
  int* s_ge = s_xe + 32;                                            // ... of g row k
  int* s_xa = s_ge + 32;                                            // the exponents the accumulators carry: columns of x channel cc ...
  int* s_ga = s_xa + 32;                                            // ... rows of g channel k (= s_xe / s_ge as of the last refresh)
  int* s_flag = s_ga + 32;                                          // [buffer]: the scan of the tile in that buffer raised an exponent
  constexpr int KSTEP_ = 256 / SPR, NG_ = 32 / KSTEP_;
  unsigned xe_p[(NLX + 3) / 4], ge_p = 0x0e0e0e0eu;                 // cached exponents (bytes) of the channels of this lane's x segments / g slots
  float scg = 1.f;                                                  // scale of this lane's g row (the A fragment's row is lane & 31)
#pragma unroll
  for (int i = 0; i < (NLX + 3) / 4; ++i) xe_p[i] = 0x0e0e0e0eu;
  static_assert(DPF_H3_EMIN == 0x0e, "packed exponent caches start at DPF_H3_EMIN");
  auto byte_of = [](unsigned w, int i) { return (int)((w >> (8 * i)) & 0xffu); };
  auto set_byte = [](unsigned& w, int i, int v) { w = (w & ~(0xffu << (8 * i))) | ((unsigned)v << (8 * i)); };
  auto seg_exp = [](const f32x4& v) {                                // biased exponent of the segment's largest magnitude (Inf / NaN: 255 -- the result is NaN either way)
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    return (int)(__builtin_bit_cast(unsigned, m) >> 23);
  };
  auto own_scan = [&](int b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's DMAs have landed; the other waves' are not read here
    const float* dbase = smem + b * bufFloats;
    int over = 0, ex[NLX], eg[NG_];                                // over: by how much a segment's exponent exceeds its channel's cached one (ONE test per scan)
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      ex[j] = 0;
      if (j * 256 < p.nxseg && xmeta[j] >= 0) {
        ex[j] = seg_exp(*reinterpret_cast<const f32x4*>(dbase + (j * 256 + tid) * 4));
        over = max(over, ex[j] - byte_of(xe_p[j >> 2], j & 3));
      }
    }
#pragma unroll
    for (int j = 0; j < NG_; ++j) {
      eg[j] = seg_exp(*reinterpret_cast<const f32x4*>(dbase + xFloats + (j * 256 + tid) * 4));
      over = max(over, eg[j] - byte_of(ge_p, j));
    }
    if (over > 0) {                                                 // (the first tiles of a workgroup, then hardly ever)
#pragma unroll
      for (int j = 0; j < NLX; ++j)
        if (j * 256 < p.nxseg && xmeta[j] >= 0 && ex[j] > byte_of(xe_p[j >> 2], j & 3)) atomicMax(&s_xe[xmeta[j] >> 24], ex[j]);
#pragma unroll
      for (int j = 0; j < NG_; ++j)
        if (eg[j] > byte_of(ge_p, j)) atomicMax(&s_ge[tid / SPR + j * KSTEP_], eg[j]);
      s_flag[b] = 1;
    }
  };
  auto pack_seg = [&](const f32x4& v, float sc) {
    unsigned h0, l0, h1, l1;
    dpf_split_pair_h(v[0] * sc, v[1] * sc, h0, l0);
    dpf_split_pair_h(v[2] * sc, v[3] * sc, h1, l1);
    u32x4 r;
    r[0] = __builtin_amdgcn_perm(h0, l0, 0x05040100u); r[1] = __builtin_amdgcn_perm(h0, l0, 0x07060302u);   // (hi << 16) | lo per element
    r[2] = __builtin_amdgcn_perm(h1, l1, 0x05040100u); r[3] = __builtin_amdgcn_perm(h1, l1, 0x07060302u);
    return r;
  };
  // (only the x patch is rewritten: an x value is read by up to 27 taps, a g value by ONE wave's two super-groups -- its split stays in
  // the unit loop, in the MFMAs' shadow, and a third of the conversion phase's traffic disappears)
  auto own_convert = [&](int b) {
    float* dbase = smem + b * bufFloats;
#pragma unroll
    for (int j = 0; j < NLX; ++j)
      if (j * 256 < p.nxseg && xmeta[j] >= 0) {
        f32x4* q = reinterpret_cast<f32x4*>(dbase + (j * 256 + tid) * 4);
        const u32x4 r = pack_seg(*q, dpf_h3_scale(byte_of(xe_p[j >> 2], j & 3)));
        *reinterpret_cast<u32x4*>(q) = r;
      }
  };
  // an exponent was raised: refresh the caches, rescale the accumulators by row and column (exact)
  auto refresh_exponents = [&]() {
#pragma unroll
    for (int j = 0; j < NLX; ++j)
      if (j * 256 < p.nxseg && xmeta[j] >= 0) set_byte(xe_p[j >> 2], j & 3, s_xe[xmeta[j] >> 24]);
#pragma unroll
    for (int j = 0; j < NG_; ++j) set_byte(ge_p, j, s_ge[tid / SPR + j * KSTEP_]);
    scg = dpf_h3_scale(s_ge[l31]);
    int dcol[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
      const int coln = t * 32 + l31, cc = (coln < ncol ? coln : 0) / p.T;
      dcol[t] = s_xa[cc] - s_xe[cc];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 oa = *reinterpret_cast<const int4*>(s_ga + 8 * q + 4 * hh), nr = *reinterpret_cast<const int4*>(s_ge + 8 * q + 4 * hh);
      const int drow[4] = {oa.x - nr.x, oa.y - nr.y, oa.z - nr.z, oa.w - nr.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[t][4 * q + r] = __builtin_ldexpf(acc[t][4 * q + r], dcol[t] + drow[r]);
    }
    __syncthreads();                                                // (workgroup-uniform branch) everyone has read the old exponents
    if (tid < 64) s_xa[tid] = s_xe[tid];                            // s_xa | s_ga <- s_xe | s_ge (adjacent pairs)
  };

  const int sw = SW1 ? 1 : p.sw;
  const int aslot = wave * 8 + hh;               // logical g slot of this lane's half, group 0
  const int arow = l31 * SPR, axor = l31 & 15;
  const long long tbeg = (long long)pchunk * p.per;
  long long tend = tbeg + p.per;
  if (tend > p.ntiles) tend = p.ntiles;
  if constexpr (X9 == 2) {
    if (tid < 128) s_xe[tid] = DPF_H3_EMIN;                        // (s_xe, s_ge, s_xa, s_ga are adjacent)
    if (tid < 2) s_flag[tid] = 0;
    __syncthreads();
  }
  if (tbeg < tend) {
    issue(0);
    if constexpr (X9 == 2) own_scan(0);
  }
  __syncthreads();
  int buf = 0;
  for (long long tile = tbeg; tile < tend; ++tile, buf ^= 1) {
    if constexpr (X9 == 2) {
      // the tile in `buf` is raw fp32 and its scan is posted: refresh the exponents if it raised one, fetch the next tile, convert this one in place
      if (__builtin_amdgcn_readfirstlane(s_flag[buf])) refresh_exponents();
      if (tile + 1 < tend) issue(buf ^ 1);
      own_convert(buf);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // conversions visible; the DMAs just issued stay in flight
      if (tid == 0) s_flag[buf] = 0;                                        // (read above by everyone; written again two tiles from now)
    } else {
      if (tile + 1 < tend) issue(buf ^ 1);
    }
    const float* s_x = smem + buf * bufFloats;
    const float* s_g = s_x + xFloats;
    // group j = positions 8j .. 8j+7 of this wave's row: lane half h takes 8j+4h .. 8j+4h+3; element i of both halves is one
    // MFMA k-step.  Units of (group, column-tile pair) are software pipelined over two B register sets; the A fragment of a
    // group is fetched one group ahead.
    if constexpr (X9 == 2) {
      const int colx = (8 - 4) * hh * sw;
      auto load_x = [&](int J, int t, unsigned (&xv)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = __builtin_bit_cast(unsigned, s_x[colbase[t] + colx + (16 * J + i) * sw]);
      };
      auto load_g = [&](int J, float (&gv)[8]) {          // raw fp32 (the g tile is not converted in place)
        const int sl0 = wave * 8 + 4 * J + 2 * hh;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(s_g + (arow + (sl0 ^ axor)) * 4);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(s_g + (arow + ((sl0 + 1) ^ axor)) * 4);
        gv[0] = g0[0]; gv[1] = g0[1]; gv[2] = g0[2]; gv[3] = g0[3]; gv[4] = g1[0]; gv[5] = g1[1]; gv[6] = g1[2]; gv[7] = g1[3];
      };
      auto split_g = [&](const float (&v)[8], f16x8& hi, f16x8& lo) {
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned a, b;
          dpf_split_pair_h(v[2 * q] * scg, v[2 * q + 1] * scg, a, b);
          h[q] = a; l[q] = b;
        }
        hi = __builtin_bit_cast(f16x8, h); lo = __builtin_bit_cast(f16x8, l);
      };
      auto unpack8 = [&](const unsigned (&v)[8], f16x8& hi, f16x8& lo) {      // 8 words (hi << 16 | lo) -> the two packed operands
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x07060302u);
          l[q] = __builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x05040100u);
        }
        hi = __builtin_bit_cast(f16x8, h); lo = __builtin_bit_cast(f16x8, l);
      };
      unsigned xr[8];
      float gv[8];
      f16x8 aH, aL, bH, bL, nH, nL;
      load_g(0, gv);
      load_x(0, 0, xr);
      split_g(gv, aH, aL);
      unpack8(xr, bH, bL);
      if (NCT > 1) load_x(0, 1, xr); else load_x(1, 0, xr);
      f16x8 a2H = aH, a2L = aL;
#pragma unroll
      for (int J = 0; J < 2; ++J) {
        if (J == 0) load_g(1, gv);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
          const int u = J * NCT + t;
          if (u + 1 < 2 * NCT) unpack8(xr, nH, nL);
          if (u + 2 < 2 * NCT) load_x((u + 2) / NCT, (u + 2) % NCT, xr);
          if (J == 0 && t == NCT - 1) split_g(gv, a2H, a2L);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aL, bH, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aH, bL, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aH, bH, acc[t], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          bH = nH; bL = nL;
        }
        aH = a2H; aL = a2L;
      }
      if (tile + 1 < tend) own_scan(buf ^ 1);          // (waits for this wave's DMAs of the next tile) its exponents, posted before the barrier
      __syncthreads();     // next tile landed, its exponents posted; this buffer is free
      continue;
    }
    if constexpr (X9 == 1) {
      // super-group J = positions 16 J .. 16 J + 15 of this wave's row: lane half h takes 16 J + 8 h .. + 7 (8 reduction indices of one
      // v_mfma_f32_32x32x16_bf16); the g fragment (two 16-byte slots) is split once per super-group, an x fragment per column tile
      const int colx = (8 - 4) * hh * sw;          // colbase[] points at position 4 hh; this path wants 8 hh
      auto load_x = [&](int J, int t, float (&xv)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = s_x[colbase[t] + colx + (16 * J + i) * sw];
      };
      auto load_g = [&](int J, float (&gv)[8]) {
        const int sl0 = wave * 8 + 4 * J + 2 * hh;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(s_g + (arow + (sl0 ^ axor)) * 4);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(s_g + (arow + ((sl0 + 1) ^ axor)) * 4);
        gv[0] = g0[0]; gv[1] = g0[1]; gv[2] = g0[2]; gv[3] = g0[3]; gv[4] = g1[0]; gv[5] = g1[1]; gv[6] = g1[2]; gv[7] = g1[3];
      };
      // software pipeline over the 2 x NCT units: the x fragment of unit u + 2 is read and the one of unit u + 1 split (vector ALU)
      // while unit u is contracted -- one MFMA, five vector instructions, alternating (sched_group_barrier)
      float xr[8], gv[8];
      bf16x8 aH, aM, aL, bH, bM, bL, nH, nM, nL;
      load_g(0, gv);
      load_x(0, 0, xr);
      split8_bf16(gv, aH, aM, aL);
      split8_bf16(xr, bH, bM, bL);
      if (NCT > 1) load_x(0, 1, xr); else load_x(1, 0, xr);
      bf16x8 a2H = aH, a2M = aM, a2L = aL;
#pragma unroll
      for (int J = 0; J < 2; ++J) {
        if (J == 0) load_g(1, gv);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
          const int u = J * NCT + t;                      // this unit; u + 1 is split now, u + 2 is read now
          if (u + 1 < 2 * NCT) split8_bf16(xr, nH, nM, nL);
          if (u + 2 < 2 * NCT) load_x((u + 2) / NCT, (u + 2) % NCT, xr);
          if (J == 0 && t == NCT - 1) split8_bf16(gv, a2H, a2M, a2L);      // g fragment of the second super-group
          if constexpr (W2_X9_FIRST <= 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aL, bL, acc[t], 0, 0, 0);
          if constexpr (W2_X9_FIRST <= 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aL, bM, acc[t], 0, 0, 0);
          if constexpr (W2_X9_FIRST <= 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aM, bL, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aL, bH, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aH, bL, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aM, bM, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aM, bH, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aH, bM, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aH, bH, acc[t], 0, 0, 0);
          {
            constexpr int NPR = 9 - W2_X9_FIRST;              // MFMAs of the unit; the split (36 vector instructions) and 8 LDS reads in their shadow
#pragma unroll
            for (int i = 0; i < NPR; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, (36 + NPR - 1) / NPR + 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, (8 + NPR - 1) / NPR, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          bH = nH; bM = nM; bL = nL;
        }
        aH = a2H; aM = a2M; aL = a2L;
      }
      __syncthreads();     // vmcnt(0): next tile landed; barrier: this buffer is free
      continue;
    }
    constexpr int NP = (NCT + 1) / 2;            // column-tile pairs per group
    f32x4 aC, aN;
    float bA[2][4], bB[2][4];
    auto load_a = [&](int j, f32x4& a) { a = *reinterpret_cast<const f32x4*>(s_g + (arow + ((aslot + 2 * j) ^ axor)) * 4); };
    auto load_b = [&](int j, int pr, float (&bb)[2][4]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * pr + u;
        if (t < NCT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bb[u][i] = s_x[colbase[t] + (8 * j + i) * sw];
        }
      }
    };
    auto mfmas = [&](int pr, const f32x4& a, const float (&bb)[2][4]) {
      if constexpr (BF) {
        const s16x4 av = pack4_bf16(a[0], a[1], a[2], a[3]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * pr + u;
          if (t < NCT) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(av, pack4_bf16(bb[u][0], bb[u][1], bb[u][2], bb[u][3]), acc[t], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int t = 2 * pr + u;
            if (t < NCT) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[u][i], acc[t], 0, 0, 0);
          }
      }
    };
    auto touch = [&](const f32x4& a, const float (&bb)[2][4], int pr) {
      asm volatile("" ::"v"(a));
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (2 * pr + u < NCT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(bb[u][i]));
        }
      asm volatile("" ::: "memory");
    };
    load_a(0, aC);
    load_b(0, 0, bA);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < 3) load_a(j + 1, aN);
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) {
        // unit (j, pr) lives in set A when (j*NP + pr) is even, else in set B; fetch the next unit into the other set first
        const int u0 = j * NP + pr;
        const int nj = pr + 1 < NP ? j : j + 1, npr = pr + 1 < NP ? pr + 1 : 0;
        if ((u0 & 1) == 0) {
          touch(aC, bA, pr);
          if (nj < 4) load_b(nj, npr, bB);
          __builtin_amdgcn_sched_barrier(6);
          mfmas(pr, aC, bA);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          touch(aC, bB, pr);
          if (nj < 4) load_b(nj, npr, bA);
          __builtin_amdgcn_sched_barrier(6);
          mfmas(pr, aC, bB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      aC = aN;
    }
    __syncthreads();     // vmcnt(0): next tile landed; barrier: this buffer is free
  }

  if constexpr (X9 == 2) {                             // back to the operands' units (exact): row and column exponents
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int coln = t * 32 + l31;
        acc[t][j] = __builtin_ldexpf(acc[t][j], s_xa[(coln < ncol ? coln : 0) / p.T] + s_ga[(j & 3) + 8 * (j >> 2) + 4 * hh] - 282);
      }
    __syncthreads();                                   // (the reduction below reuses this LDS)
  }
  // ---- sum the four waves' partial tiles through LDS (two rounds: 2,3 -> 0,1 then 1 -> 0); image [tile][reg][lane], conflict free
  float* red = smem;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int src_lo = round == 0 ? 2 : 1;       // waves in [src_lo, 2*src_lo) write, waves < src_lo (partner = wave + src_lo) add
    if (wave >= src_lo && wave < 2 * src_lo) {
      float* dst = red + (wave - src_lo) * (NCT * 16 * 64) + lane;
#pragma unroll
      for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) dst[(t * 16 + j) * 64] = acc[t][j];
    }
    __syncthreads();
    if (wave < src_lo) {
      const float* src = red + wave * (NCT * 16 * 64) + lane;
#pragma unroll
      for (int t = 0; t < NCT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] += src[(t * 16 + j) * 64];
    }
    __syncthreads();
  }
  // ---- wave 0: partial dW -> slab[pchunk][kbase + k][c0*T + coln]   (plain stores; summed by wgrad2_reduce_kernel)
  if (wave == 0) {
    float* sl = slab + (long long)pchunk * p.slab_stride + (long long)kbase * p.C * p.T + (long long)c0 * p.T;
    const long long rowlen = (long long)p.C * p.T;
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
      const int coln = t * 32 + l31;
      if (coln < ncol) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = (j & 3) + 8 * (j >> 2) + 4 * hh;
          if (k < krows) sl[(long long)k * rowlen + coln] = acc[t][j];
        }
      }
    }
  }
}

// dw[i] (+)= sum_{chunk} slab[chunk][i] in a fixed order (deterministic), in two levels so that the fold is not one serial chain of
// nchunk dependent loads per thread on a quarter-filled chip: grid.y groups of W2_RG chunks -> part[group][i], then the groups
constexpr int W2_RG = 16;
__global__ void wgrad2_fold_kernel(const float* __restrict__ slab, float* __restrict__ part, long long n, int nchunk) {
  const int c0 = blockIdx.y * W2_RG, c1 = c0 + W2_RG < nchunk ? c0 + W2_RG : nchunk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = c0; c < c1; ++c) s += slab[(long long)c * n + i];
    part[(long long)blockIdx.y * n + i] = s;
  }
}
__global__ void wgrad2_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int nchunk, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += slab[(long long)c * n + i];
    dw[i] = accumulate ? dw[i] + s : s;
  }
}

unsigned magic20(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }
int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// bank conflicts of the B-operand gather for one candidate (PS, CS): sum over the 32-lane column windows of the extra LDS cycles
int gather_conflicts(const DpfWgradDesc& d, int T, int ncolmax, int RS, int PS, int CS, int dhl) {
  int total = 0;
  for (int c0 = 0; c0 < ncolmax; c0 += 32) {
    int cnt[32] = {0};
    for (int l = 0; l < 32 && c0 + l < ncolmax; ++l) {
      const int col = c0 + l, cc = col / T, tap = col % T;
      const int tw = tap % d.kw, th = (tap / d.kw) % d.kh, td = tap / (d.kw * d.kh);
      ++cnt[(cc * CS + td * d.dd * PS + th * dhl * RS + tw * d.dw) & 31];
    }
    int worst = 0;
    for (int b = 0; b < 32; ++b) worst = cnt[b] > worst ? cnt[b] : worst;
    total += worst - 1;
  }
  return total;
}

template <int NCT, bool BF, bool SW1, int X9>
int launch_w2c(const float* g, const float* x, float* slab, const W2P& p, size_t lds, unsigned blocks, hipStream_t st) {
  static bool done = false;
  if (lds > 48 * 1024 && !done) {
    if (hipFuncSetAttribute((const void*)wgrad2_kernel<NCT, BF, SW1, X9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return DPF_ERR_LAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((wgrad2_kernel<NCT, BF, SW1, X9>), dim3(blocks), dim3(256), lds, st, g, x, slab, p);
  return dpf_check_launch();
}
template <int NCT, bool BF, int X9>
int launch_w2b(const float* g, const float* x, float* slab, const W2P& p, size_t lds, unsigned blocks, hipStream_t st) {
  static const int sw1 = env_int("DPF_W2_SW1", 1);
  return (p.sw == 1 && sw1) ? launch_w2c<NCT, BF, true, X9>(g, x, slab, p, lds, blocks, st) : launch_w2c<NCT, BF, false, X9>(g, x, slab, p, lds, blocks, st);
}
template <int NCT>
int launch_w2(const float* g, const float* x, float* slab, const W2P& p, size_t lds, unsigned blocks, hipStream_t st) {
  if (dpf_conv_operand_bf16()) return launch_w2b<NCT, true, 0>(g, x, slab, p, lds, blocks, st);
  // f16 components: the in-place conversion of a tile is a fixed cost per tile that only tiles with many units repay (stride-2 launches
  // -- two column tiles, a 2.8x larger patch per channel -- ran 40 % slower with it; with only the x patch converted five column tiles gain
  // 15 %, four break even): NCT >= 5, else six bf16 products
  switch (dpf_conv_f32_x9() == 2 && NCT < 5 ? 1 : dpf_conv_f32_x9()) {
    case 2: return launch_w2b<NCT, false, 2>(g, x, slab, p, lds, blocks, st);
    case 1: return launch_w2b<NCT, false, 1>(g, x, slab, p, lds, blocks, st);
    default: return launch_w2b<NCT, false, 0>(g, x, slab, p, lds, blocks, st);
  }
}

int w2_maxblocks() { return 1024; }     // upper bound of resident workgroups (4 per CU): sizes the slab workspace

}  // namespace

long long dpf_wgrad2_workspace_floats(int T, int C, int K) {
  // position chunks <= maxblocks / (column chunks * k slices), column chunks >= ceil(C*T / 224): slabs <= maxblocks * 224 * 32 floats (+ rounding)
  const int Kc = K < 128 ? K : 128;
  const long long cols = (long long)C * T;
  const long long groups = ((cols + 223) / 224) * ((Kc + 31) / 32);
  const long long nchunk = w2_maxblocks() / groups + 8;
  return (nchunk + nchunk / 16 + 2) * Kc * cols + 64;          // slabs + the group sums of the two-level fold
}

int dpf_wgrad2(const float* g, const float* x, float* dw, float* ws, long long ws_floats, const DpfWgradDesc& d, int accumulate, hipStream_t st) {
  static const int enabled = env_int("DPF_WGRAD2", 1);
  if (!enabled || !ws) return DPF_ERR_UNSUPPORTED;
  const int T = d.kd * d.kh * d.kw;
  constexpr int min_t = 9;
  if (T > 27 || T < min_t || d.K > 128) return DPF_ERR_UNSUPPORTED;
  if ((d.IW & 3) || (d.QW & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(g) & 15)) return DPF_ERR_UNSUPPORTED;
  constexpr int maxdil = 8;
  if (d.dh > maxdil || d.dw > maxdil || d.dd > maxdil) return DPF_ERR_UNSUPPORTED;   // widely dilated: polyphase kernel (conv_igemm.hip)
  if (d.sw > 2 || d.sh > 2) return DPF_ERR_UNSUPPORTED;
  const long long x_chan = (long long)d.ID * d.IH * d.IW;
  constexpr int nct_max = 7, lds_max = 80 * 1024;

  W2P p{};
  p.N = d.N; p.C = d.C; p.K = d.K; p.Ktot = d.Ktot; p.k0 = d.k0;
  p.ID = d.ID; p.IH = d.IH; p.IW = d.IW; p.QD = d.QD; p.QH = d.QH; p.QW = d.QW;
  p.sd = d.sd; p.sh = d.sh; p.sw = d.sw; p.pd = d.pd; p.ph = d.ph; p.pw = d.pw;
  p.kd = d.kd; p.kh = d.kh; p.kw = d.kw; p.dd = d.dd; p.dh = d.dh; p.dw = d.dw; p.T = T;
  // rows dilated by dh at stride 1: a tile takes the 4 rows of ONE dilation phase (dh apart), so its patch is kh + 3 image rows instead
  // of 4 + (kh - 1) * dh (20 rows at dilation 8: a fifth of the channels per LDS buffer, 44.7 TFLOP/s on the 64 -> 64 dilation-8 layer)
  static const int rstep_on = env_int("DPF_W2_RSTEP", 1);
  p.rstep = (rstep_on && d.sh == 1 && d.dh > 1 && d.kh > 1) ? d.dh : 1;
  p.dhl = d.dh / p.rstep;
  p.ext_d = (d.kd - 1) * d.dd + 1; p.ext_h = (WTH - 1) * d.sh + (d.kh - 1) * p.dhl + 1;
  const int ext_w = 31 * d.sw + (d.kw - 1) * d.dw + 1;
  p.colshift = (((-d.pw) % 4) + 4) % 4;
  p.RS = ((p.colshift + ext_w + 3) / 4) * 4;
  p.SR = p.RS / 4;
  // column chunks: whole channels, balanced, at most nct_max tiles of 32 columns; shrink until the two LDS buffers fit
  int NCT = 0, CCW = 0;
  // stride 2: the x patch of a tile is 2.8x larger per channel; two column tiles (fewer channels per buffer, more resident
  // workgroups) measured 69 vs 56 TFLOP/s against the stride-1 optimum of seven
  const int nct_cap = (d.sh == 2 || d.sw == 2) ? 2 : nct_max;
  for (int nmax = nct_cap; nmax >= 1; --nmax) {
    int ccw_cap = (nmax * 32) / T;
    if (ccw_cap < 1) continue;
    const int cchunks = dpf_div_up(d.C, ccw_cap);
    CCW = dpf_div_up(d.C, cchunks);
    NCT = dpf_div_up(CCW * T, 32);
    // pad the plane / channel strides (multiples of 4 floats) for the fewest gather conflicts
    int bestc = 1 << 30;
    for (int pp = 0; pp < 8; ++pp)
      for (int cp = 0; cp < 8; ++cp) {
        const int PS = p.ext_h * p.RS + 4 * pp, CS = p.ext_d * PS + 4 * cp;
        const int c = gather_conflicts(d, T, CCW * T, p.RS, PS, CS, p.dhl) * 64 + pp * p.ext_d + cp;
        if (c < bestc) { bestc = c; p.PS = PS; p.CS = CS; }
      }
    const size_t buf = (size_t)(CCW * p.CS + GFLOATS) * sizeof(float);
    const size_t red = (size_t)2 * NCT * 16 * 64 * sizeof(float);
    const size_t lds = (2 * buf > red ? 2 * buf : red) + 576;   // + the exponent tables of the f16-component path
    if (CCW * p.CS / 4 > NLX * 256 || lds > (size_t)lds_max) { NCT = 0; continue; }
    break;
  }
  if (NCT == 0) return DPF_ERR_UNSUPPORTED;
  if (9LL * (CCW + 1) * x_chan >= (1LL << 30) || 33LL * (long long)d.QD * d.QH * d.QW >= (1LL << 31)) return DPF_ERR_UNSUPPORTED;
  if ((long long)NLX * 256 * (p.CS / 4) >= (1LL << 20)) return DPF_ERR_UNSUPPORTED;
  p.PSseg = p.PS / 4; p.CSseg = p.CS / 4;
  p.CCW = CCW; p.nxseg = CCW * p.CS / 4;
  p.cchunks = dpf_div_up(d.C, CCW);
  p.kslices = dpf_div_up(d.K, 32);
  p.tilesH = dpf_div_up(d.QH, WTH * p.rstep) * p.rstep; p.tilesW = dpf_div_up(d.QW, 32);
  p.ntiles = (long long)d.N * d.QD * p.tilesH * p.tilesW;
  p.mCS = magic20(p.CSseg); p.mPS = magic20(p.PSseg); p.mSR = magic20(p.SR);
  // as many position chunks as fit the chip at once: (workgroups resident per CU by registers and LDS) x 256 CUs / groups
  p.groups = p.cchunks * p.kslices;
  const size_t buf0 = (size_t)(CCW * p.CS + GFLOATS) * sizeof(float);
  const size_t red0 = (size_t)2 * NCT * 16 * 64 * sizeof(float);
  const size_t lds0 = (2 * buf0 > red0 ? 2 * buf0 : red0) + 576;
  int occ = w2_occ_of(NCT);
  if ((size_t)occ * lds0 > 160 * 1024) occ = (int)((160 * 1024) / lds0);
  if (occ < 1) occ = 1;
  constexpr int cap_over = 0;
  // one resident round of workgroups; with many (column chunk, k slice) groups the position chunks get coarse, and two rounds
  // balance the CUs better (measured: K = 81 / 96 layers 58 -> 71-86 TFLOP/s, the 32-channel layers unchanged)
  const int capacity = cap_over ? cap_over : occ * 256 * (p.groups >= 12 ? 2 : 1);
  long long nchunk = capacity / p.groups;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > p.ntiles) nchunk = p.ntiles;
  // position chunks in multiples of 16 (8 XCD labels x 2): the workgroups of a chunk share their x / g tiles through one XCD's L2, and
  // an uneven deal of chunks to XCDs costs 15-25 % (K = 81 layers: 42 chunks 73 TFLOP/s, 32 chunks 89; 68 chunks 73, 64 chunks 86)
  if (!cap_over) {
    if (nchunk >= 16 && (nchunk & 7)) nchunk &= ~15LL;          // already a multiple of 8: keep (56 chunks beat 48 on the 96-channel layer)
    else if (nchunk > 8 && nchunk < 16) nchunk = 8;
  }
  p.per = (p.ntiles + nchunk - 1) / nchunk;
  nchunk = (p.ntiles + p.per - 1) / p.per;
  p.nchunk = (int)nchunk;
  p.slab_stride = (long long)d.K * d.C * T;
  if (nchunk * p.slab_stride > ws_floats) return DPF_ERR_UNSUPPORTED;

  const size_t buf = (size_t)(CCW * p.CS + GFLOATS) * sizeof(float);
  const size_t red = (size_t)2 * NCT * 16 * 64 * sizeof(float);
  const size_t lds = (2 * buf > red ? 2 * buf : red) + 576;
  const unsigned blocks = (unsigned)(8 * ((p.nchunk + 7) / 8) * p.groups);
  int rc;
  switch (NCT) {
    case 1: rc = launch_w2<1>(g, x, ws, p, lds, blocks, st); break;
    case 2: rc = launch_w2<2>(g, x, ws, p, lds, blocks, st); break;
    case 3: rc = launch_w2<3>(g, x, ws, p, lds, blocks, st); break;
    case 4: rc = launch_w2<4>(g, x, ws, p, lds, blocks, st); break;
    case 5: rc = launch_w2<5>(g, x, ws, p, lds, blocks, st); break;
    case 6: rc = launch_w2<6>(g, x, ws, p, lds, blocks, st); break;
    default: rc = launch_w2<7>(g, x, ws, p, lds, blocks, st); break;
  }
  if (rc != DPF_OK) return rc;
  const long long n = p.slab_stride;
  const int groups2 = (p.nchunk + W2_RG - 1) / W2_RG;
  if (p.nchunk > 2 * W2_RG && (nchunk + groups2) * n <= ws_floats) {
    float* part = ws + nchunk * n;
    hipLaunchKernelGGL(wgrad2_fold_kernel, dim3(dpf_ew_grid(n), groups2), dim3(256), 0, st, ws, part, n, p.nchunk);
    hipLaunchKernelGGL(wgrad2_reduce_kernel, dim3(dpf_ew_grid(n)), dim3(256), 0, st, part, dw, n, groups2, accumulate);
  } else {
    hipLaunchKernelGGL(wgrad2_reduce_kernel, dim3(dpf_ew_grid(n)), dim3(256), 0, st, ws, dw, n, p.nchunk, accumulate);
  }
  return dpf_check_launch();
}
