// Dense convolution as implicit GEMM on the fp32 matrix cores, second generation: LDS-DMA double buffering.
//
// Same math and MFMA mapping as conv_igemm.hip (D[row = out channel][col = 32 consecutive W positions], exact-f32
// v_mfma_f32_32x32x2_f32, one 256-thread workgroup owns ALL output channels of a (4*NT) x 32 position tile), but the
// operand traffic is re-designed around what bounds that kernel on MI355X (profiles/r01_mfma_probe.txt: the matrix pipe
// idles ~50 % behind the stage -> barrier -> MFMA -> barrier rhythm and the in-order vmcnt queue shared by weight and
// patch loads):
//   * every chunk of CC input channels is fetched by `global_load_lds_dwordx4` (LDS-DMA, 16 B per lane, no VGPR round trip,
//     no ds_write): the haloed input patch as whole 16-byte-aligned row segments (per-lane SOURCE addresses; out-of-bounds
//     segments read a zero page, so padding needs no predication in the MFMA loop), and the chunk's weights, which the
//     repack lays out [chunk][tap][cc][k] so a chunk is one contiguous block;
//   * two LDS buffers: the DMA of chunk i+1 is issued before the MFMA loop of chunk i and retired by the one
//     vmcnt(0) + barrier at the end of that loop -- no global load result is ever waited for inside the MFMA stream, and
//     both MFMA operands come from LDS (the weight fragments no longer share the vmcnt queue with the patch);
//   * descriptors (per-lane source offsets) are computed once per workgroup with multiply-shift divisions.
// Eligibility (host): IW % 4 == 0, 16-byte aligned x, forward convs of any stride / dilation and stride-1 transposed
// convs (data gradients); everything else stays on conv_igemm.hip.
//
// Kernels of this file (one host entry, dpf_igemm2_conv, picks):
//   igemm3_x9_kernel   stride-1 launches (forward and transposed): fp32 products from exact bf16 splits on the bf16 matrix pipe, or
//                      bf16-rounded operands (operand precision "bf16"); patch staged through registers, split once per element
//   igemm2_kernel      every other eligible launch on v_mfma_f32_32x32x2_f32 (stride 2, patches beyond the x9 staging budget, C < 8),
//                      and its bf16-operand variant
//   igemm2_tr2_kernel  stride-2 transposed 3x3x3 (class-fused)
// They share the tile geometry (G2P), the XCD-aware tile order and the tile epilogue (g2_epilogue: bias, accumulate, 16-byte stores,
// fused BatchNorm statistics).
#include "conv_internal.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int MAXT = 27;
constexpr int NLD = 8;          // patch DMA instructions per thread and chunk (upper bound; 2048 x 16 B = 32 KB per chunk)
constexpr int ZPAGE = 16;       // floats of zeros in front of the packed weights (source of out-of-bounds segments)

struct G2P {
  int N, C, K, Ktot, k0;
  int ID, IH, IW, OD, OH, OW;
  int sxd, sxh, sxw;            // input step per output step
  int e0d, e0h, e0w;            // input coordinate of output 0 at the minimal tap: i = q * sx + e0
  int ext_d, ext_h;             // patch planes, rows per plane
  int pz, thp, thp_shift, odt;  // a tile = pz output planes x thp rows (pz * thp = 4 * NT position rows); odt = depth tiles
  int planeStride;              // floats between patch planes (ext_h * RS)
  int RS, SR;                   // patch row stride in floats (multiple of 4), 16-byte segments per row
  int colshift;                 // patch column of the first needed input column (alignment slack, 0..3)
  int rpc, chanStride;          // rows per channel, floats per channel
  int nseg;                     // 16-byte segments of one chunk's patch  (CC * rpc * SR)
  int nwseg;                    // 16-byte segments of one chunk's weights (T * CC * KT / 4)
  int T;
  int tilesH, tilesW;
  int ntiles, cpx;              // tiles; tiles per XCD label
  int nchunks;
  unsigned mSR, mRPC, mEH;      // ceil(2^20 / d) for d = SR, rpc, ext_h
  unsigned long long steps;     // 2-bit step code per tap (see the kernel's tap walk)
  int rstep;                    // rows of a tile are rstep apart (1; the x9 kernel's 2-D dilated layers: the rows of ONE dilation phase)
  int accum;                    // 1: out += result
  int vec;                      // 1: OW % 4 == 0 and a 16-byte aligned output: 16-byte stores (the epilogue transposes 4 x 4 blocks across lanes)
  int single;                   // 1: one LDS buffer (more resident workgroups hide the DMA instead of a second buffer)
  double* stats;                // optional [ntiles][statsK][2] per-tile (sum, sum of squares) of the outputs, for a following BatchNorm
  int statsK, statsk0;          // row length of the slab and first channel of this launch in it (K, 0 unless the launch is one slice of the channels)
  int tapoff[28];               // x9 kernel: patch offset (positions) of tap u; taps beyond T: 0 (their weights are zero)
  const int* wexp;              // x9 kernel, f16 components: [32 MT] biased exponent the pack kernel scaled output row k's weights by (device memory)
  int guard;                    // x9 kernel, f16 components: residual passes for chunks whose range exceeds the split's (always 1 outside tests)
  int tap0, stepC, incB, incA;   //   incB = stepB - (kw-1)*stepC (row wrap), incA = stepA - (kh-1)*stepB - (kw-1)*stepC (plane wrap) // patch offset (floats) of tap (a, b, c) = tap0 + a*stepA + b*stepB + c*stepC
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two floats -> two bf16 (round to nearest even) in one register: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

__device__ __forceinline__ void glds16(const float* gsrc, float* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

// Tile epilogue shared by the tile kernels of this file (accumulator layout of the 32x32 MFMAs, f32 or bf16 operands alike):
// D row = (j&3) + 8*(j>>2) + 4*(lane>>5), col = lane&31.
template <int MT, int NT>
__device__ __forceinline__ void g2_epilogue(f32x16 (&acc)[MT][NT], const G2P& p, const float* __restrict__ bias, float* __restrict__ out, float* smem,
                                            int tile_id, int n, int qd, int q0h, int q0w, int wave, int l31, int hh, int tid) {
  constexpr int KT = 32 * MT;
  const int ow = q0w + l31;
  const long long out_plane = (long long)p.OH * p.OW;
  const long long kstride = (long long)p.OD * out_plane;
  float* op = out + ((long long)n * p.Ktot + p.k0) * kstride + ow;
  long long orow[NT];      // offset of position row t of this wave inside a channel
  bool rok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = wave * NT + t, oz = qd + (r >> p.thp_shift), oy = q0h + (r & (p.thp - 1)) * p.rstep;
    rok[t] = oz < p.OD && oy < p.OH;
    orow[t] = ((long long)oz * p.OH + oy) * p.OW;
  }
  if (p.vec) {
    // 16-byte stores.  A lane holds ONE column and 16 rows of a 32 x 32 tile, so plain stores are 4 bytes per lane and 64 instructions
    // per thread -- the store tail then takes as long as the MFMA loop of a 32-channel 2-D layer (in-kernel stamps: 30 k of 72 k clocks).
    // Each 4 x 4 block (rows 8 i + 4 hh + 0..3, columns 4 c .. 4 c + 3: registers 4 i .. 4 i + 3 of the four lanes 4 c .. 4 c + 3) is
    // transposed across its quad with two DPP exchange stages; lane 4 c + q then owns row q of the block, four consecutive columns.
    const int q = l31 & 3, owv = q0w + (l31 & ~3);
    const bool o1 = (q & 1) != 0, o2 = (q & 2) != 0;
    float* opv = out + ((long long)n * p.Ktot + p.k0) * kstride + owv;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = m * 32 + 8 * i + 4 * hh + q;
        const float bv = (bias && k < p.K) ? bias[p.k0 + k] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float r0 = acc[m][t][4 * i], r1 = acc[m][t][4 * i + 1], r2 = acc[m][t][4 * i + 2], r3 = acc[m][t][4 * i + 3];
          {  // lanes q ^ 1: quad_perm [1, 0, 3, 2]
            const float x = o1 ? r0 : r1, y = o1 ? r2 : r3;
            const float xs = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
            const float ys = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0xB1, 0xf, 0xf, true));
            if (o1) { r0 = xs; r2 = ys; } else { r1 = xs; r3 = ys; }
          }
          {  // lanes q ^ 2: quad_perm [2, 3, 0, 1]
            const float x = o2 ? r0 : r2, y = o2 ? r1 : r3;
            const float xs = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
            const float ys = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0x4E, 0xf, 0xf, true));
            if (o2) { r0 = xs; r1 = ys; } else { r2 = xs; r3 = ys; }
          }
          if (rok[t] && k < p.K && owv < p.OW) {
            f32x4* o = reinterpret_cast<f32x4*>(opv + (long long)k * kstride + orow[t]);
            f32x4 v = {r0 + bv, r1 + bv, r2 + bv, r3 + bv};
            if (p.accum) v += *o;
            *o = v;
          }
        }
      }
  } else if (ow < p.OW && !p.accum) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (k < p.K) {
          const float bv = bias ? bias[p.k0 + k] : 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (rok[t]) op[(long long)k * kstride + orow[t]] = acc[m][t][j] + bv;
        }
      }
    }
  } else if (ow < p.OW) {                                         // out += result (kept apart: the plain path has no load in its store loop)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (k < p.K) {
          const float bv = bias ? bias[p.k0 + k] : 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (rok[t]) {
              float* o = op + (long long)k * kstride + orow[t];
              *o += acc[m][t][j] + bv;
            }
        }
      }
    }
  }
  // ---- optional BatchNorm statistics of this tile (the consumer's bn_stats pass over the output tensor is then not needed):
  // per output channel the sum and the sum of squares of the valid outputs, accumulated in fp64 from the first add, reduced over the 32 columns by shuffles and over the 4 waves through LDS in a fixed order, one
  // [K][2] row per tile.  A finalize kernel folds the rows in a fixed order: deterministic, no atomics, no zero fill.
  if (p.stats) {                                                 // uniform
    double* red = reinterpret_cast<double*>(smem);               // [4 waves][2][MT * 16][2]; the operand buffers are dead by now
    const bool colok = ow < p.OW;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      double v1[16], v2[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = m * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
        const float bv = (bias && k < p.K) ? bias[p.k0 + k] : 0.f;
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const double v = (colok && rok[t]) ? (double)(acc[m][t][j] + bv) : 0.0;
          d1 += v;
          d2 += v * v;                                           // exact product, fp64 accumulation
        }
        v1[j] = d1;
        v2[j] = d2;
      }
      // transpose-reduce over the 32 columns: at every step a lane hands half of its remaining rows to its partner and adds the
      // partner's half of the rows it keeps -- 8 + 4 + 2 + 1 exchanges instead of 16 x 4, then one exchange between the two
      // lanes that ended up with the same row.  Fixed order.
#pragma unroll
      for (int half = 8; half >= 1; half >>= 1) {
        const bool upper = (l31 & (2 * half)) != 0;              // lane bit 4, 3, 2, 1 for half = 8, 4, 2, 1
#pragma unroll
        for (int i = 0; i < half; ++i) {
          const double s1 = upper ? v1[i] : v1[i + half], s2 = upper ? v2[i] : v2[i + half];
          const double k1 = upper ? v1[i + half] : v1[i], k2 = upper ? v2[i + half] : v2[i];
          v1[i] = k1 + __shfl_xor(s1, 2 * half, 64);
          v2[i] = k2 + __shfl_xor(s2, 2 * half, 64);
        }
      }
      const double r1 = v1[0] + __shfl_xor(v1[0], 1, 64), r2 = v2[0] + __shfl_xor(v2[0], 1, 64);
      if ((l31 & 1) == 0) {
        const int j = (l31 >> 1) & 15;                           // bit 4 -> +8, bit 3 -> +4, bit 2 -> +2, bit 1 -> +1
        double* dst = red + (((wave * 2 + hh) * MT + m) * 16 + j) * 2;
        dst[0] = r1;
        dst[1] = r2;
      }
    }
    __syncthreads();
    if (tid < KT && tid < p.K) {
      const int m = tid >> 5, kk = tid & 31;
      const int h2 = (kk >> 2) & 1, j = (kk & 3) + 4 * (kk >> 3);
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        const double* src = red + (((wv * 2 + h2) * MT + m) * 16 + j) * 2;
        a1 += src[0];
        a2 += src[1];
      }
      double* row = p.stats + ((long long)tile_id * p.statsK + p.statsk0 + tid) * 2;
      row[0] = a1;
      row[1] = a2;
    }
  }
}

// resident workgroups per CU the register budget is declared for: accumulators + two operand sets + ~40 of bookkeeping
template <int MT, int NT, int CC>
constexpr int g2_occ() {
  constexpr int est = MT * NT * 16 + CC * (MT + NT) + 40;
  return est <= 128 ? 4 : (est <= 168 ? 3 : 2);
}

// BF = true (operand precision "bf16"): same staging (fp32 patch by LDS-DMA), but a chunk is 8 channels, the weights are packed as
// bf16 [tap][half][k][4], and a tap is ONE v_mfma_f32_32x32x8_bf16 per (row tile, position row): a lane reads its 4 channels of the
// fp32 patch, rounds them to bf16 (RNE) and contracts 8 channels at once -- 1/8 of the matrix-pipe cycles of the exact-f32 path,
// fp32 accumulation, fp32 output.  The result equals an fp32 convolution of the bf16-rounded operands up to summation order.
template <int MT, int NT>
constexpr int g2_occ_bf() {
  constexpr int est = MT * NT * 16 + 2 * (2 * MT + 4 * NT) + 2 * NLD + 44;
  return est <= 128 ? 4 : (est <= 168 ? 3 : 2);
}

template <int MT, int NT, int CC, bool BF = false>
__global__ __launch_bounds__(256, (BF ? g2_occ_bf<MT, NT>() : g2_occ<MT, NT, CC>())) void igemm2_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                         const float* __restrict__ bias, float* __restrict__ out, G2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(!BF || CC == 8, "the bf16 path contracts 8 channels per MFMA");
  constexpr int KT = 32 * MT;
  constexpr int TH = 4 * NT;
  constexpr int NLDK = BF ? 2 * NLD : NLD;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hh = lane >> 5;

  // L2-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (b % 8 labels the XCD), so each label walks a
  // contiguous tile range ordered depth-fastest -- the workgroups that share input planes and halo rows run together behind one L2
  int b = (blockIdx.x & 7) * p.cpx + (blockIdx.x >> 3);
  if (b >= p.ntiles) return;
  const int tile_id = b;
  const int qd = (b % p.odt) * p.pz; b /= p.odt;       // first output plane of the tile
  const int tw = b % p.tilesW; b /= p.tilesW;
  const int th = b % p.tilesH;
  const int n = b / p.tilesH;
  const int q0h = th * p.thp, q0w = tw * 32;
  const int i0d = qd * p.sxd + p.e0d, i0h = q0h * p.sxh + p.e0h;
  const int a0 = q0w * p.sxw + p.e0w - p.colshift;     // 16-byte aligned first staged column

  const int patchFloats = CC * p.chanStride;
  const int bufFloats = patchFloats + 4 * p.nwseg;
  const long long x_chan = (long long)p.ID * p.IH * p.IW;
  const float* xn = x + (long long)n * p.C * x_chan;

  // ---- per-lane DMA descriptors: segment f = tid + 256 j of the flat [row][SR] patch image
  int goff[NLDK];
#pragma unroll
  for (int j = 0; j < NLDK; ++j) {
    const unsigned f = tid + 256 * j;
    const unsigned row = (f * p.mSR) >> 20;
    const int seg = f - row * p.SR;
    const unsigned cc = (row * p.mRPC) >> 20;
    const unsigned rem = row - cc * p.rpc;
    const unsigned pl = (rem * p.mEH) >> 20;
    const int rr = rem - pl * p.ext_h;
    const int id = i0d + (int)pl, ih = i0h + rr, iw = a0 + 4 * seg;
    const bool ok = (int)f < p.nseg && id >= 0 && id < p.ID && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
    goff[j] = ok ? (int)((long long)cc * x_chan + ((long long)id * p.IH + ih) * p.IW + iw) : -1;
  }

  auto issue = [&](int chunk, int buf) {
    float* dbase = smem + buf * bufFloats;
    const float* xc = xn + (long long)chunk * CC * x_chan;
    const int crem = p.C - chunk * CC;
    const int flimit = (crem < CC ? crem : CC) * p.rpc * p.SR;    // segments of channels beyond C read the zero page
#pragma unroll
    for (int j = 0; j < NLDK; ++j) {
      if (j * 256 < p.nseg) {                                     // wave-uniform
        const int f = tid + 256 * j;
        if (f < p.nseg) {
          const float* src = (goff[j] >= 0 && f < flimit) ? xc + goff[j] : wpk;
          glds16(src, dbase + (j * 256 + wave * 64) * 4);
        }
      }
    }
    const float* wc = wpk + ZPAGE + (long long)chunk * p.nwseg * 4;
    float* wbase = dbase + patchFloats;
    for (int f0 = wave * 64; f0 < p.nwseg; f0 += 256) {
      const int f = f0 + lane;
      if (f < p.nwseg) glds16(wc + f * 4, wbase + f0 * 4);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;

  int lanebase[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    // position row r of the tile = (plane r / thp, row r % thp): depth tiles share the kd - 1 halo planes between their output planes
    const int r = wave * NT + t, rz = r >> p.thp_shift, ry = r & (p.thp - 1);
    lanebase[t] = rz * p.sxd * p.planeStride + ry * p.sxh * p.RS + l31 * p.sxw + p.colshift + hh * (BF ? 4 : 1) * p.chanStride;
  }
  const int abase = hh * KT + l31;

  issue(0, 0);
  __syncthreads();                                               // vmcnt(0) + barrier: chunk 0 landed
  const int nv = p.T;
  const int tap0 = p.tap0, stepC = p.stepC, dB = p.incB - p.stepC, dA = p.incA - p.incB, chanStride = p.chanStride;
  const unsigned long long steps = p.steps;
  const int single = p.single;
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int buf = single ? 0 : chunk & 1;
    if (!single && chunk + 1 < p.nchunks) issue(chunk + 1, buf ^ 1);
    const float* s_in = smem + buf * bufFloats;
    const float* s_w = s_in + patchFloats;
    if constexpr (BF) {
      // bf16 operands: per tap a lane fetches its 4 channels of the fp32 patch (4 ds_read_b32) and its 4 packed bf16 weights per
      // row tile (one ds_read_b64); same two-set software pipeline and scalar tap walk as the exact-f32 path below.
      const short* s_wb = reinterpret_cast<const short*>(s_w) + (hh * KT + l31) * 4;
      s16x4 aA[MT], aB[MT];
      float bA[NT][4], bB[NT][4];
      int slotn = 0, toff = tap0, woff = 0;
      auto load_ops = [&](s16x4 (&a)[MT], float (&bb)[NT][4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const s16x4*>(s_wb + woff + m * 128);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) bb[t][i] = s_in[lanebase[t] + i * chanStride + toff];
        const unsigned code = (unsigned)(steps >> (2 * slotn)) & 3u;
        const int inc = stepC + (code > 0 ? dB : 0) + (code > 1 ? dA : 0);
        toff = code == 3 ? tap0 : toff + inc;
        woff = code == 3 ? 0 : woff + 8 * KT;
        slotn = code == 3 ? 0 : slotn + 1;
      };
      auto mfmas = [&](const s16x4 (&a)[MT], const float (&bb)[NT][4]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const u32x2 u = {pk_bf16(bb[t][0], bb[t][1]), pk_bf16(bb[t][2], bb[t][3])};
          const s16x4 bv = __builtin_bit_cast(s16x4, u);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[m], bv, acc[m][t], 0, 0, 0);
        }
      };
      auto touch = [&](const s16x4 (&a)[MT], const float (&bb)[NT][4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(a[m]));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(bb[t][i]));
        asm volatile("" ::: "memory");
      };
      load_ops(aA, bA);
      for (int slot = 0; slot + 1 < nv; slot += 2) {
        touch(aA, bA);
        load_ops(aB, bB);
        __builtin_amdgcn_sched_barrier(6);
        mfmas(aA, bA);
        __builtin_amdgcn_sched_barrier(0);
        touch(aB, bB);
        load_ops(aA, bA);
        __builtin_amdgcn_sched_barrier(6);
        mfmas(aB, bB);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nv & 1) mfmas(aA, bA);
    } else {
      // Two operand register sets (A, B), taps two at a time: the LDS reads of tap s+1 are issued before the MFMAs of tap s, so
      // the matrix pipe never waits for an operand fetch.  Tap offsets advance incrementally on the scalar unit
      // (tap (a, b, c) -> base + a*stepA + b*stepB + c*stepC; weights are packed in the same order).
      float aA[CC / 2][MT], bA[CC / 2][NT], aB[CC / 2][MT], bB[CC / 2][NT];
      int slotn = 0, toff = tap0, woff = 0;
      auto load_ops = [&](float (&a)[CC / 2][MT], float (&bb)[CC / 2][NT]) {
        const float* wrow = s_w + abase + woff;
  #pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp) {
  #pragma unroll
          for (int m = 0; m < MT; ++m) a[cp][m] = wrow[(2 * cp) * KT + m * 32];
  #pragma unroll
          for (int t = 0; t < NT; ++t) bb[cp][t] = s_in[lanebase[t] + (2 * cp) * chanStride + toff];
        }
        // advance to the next tap on the scalar unit: 2-bit step codes, one per tap (0: next column, 1: next row, 2: next plane,
        // 3: wrap to tap 0 -- the wrapped fetch of the final iteration is a valid address whose data is not used)
        const unsigned code = (unsigned)(steps >> (2 * slotn)) & 3u;
        const int inc = stepC + (code > 0 ? dB : 0) + (code > 1 ? dA : 0);     // arithmetic, not a select of kernel-argument loads
        toff = code == 3 ? tap0 : toff + inc;
        woff = code == 3 ? 0 : woff + CC * KT;
        slotn = code == 3 ? 0 : slotn + 1;
      };
      auto mfmas = [&](const float (&a)[CC / 2][MT], const float (&bb)[CC / 2][NT]) {
  #pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp)
  #pragma unroll
          for (int m = 0; m < MT; ++m)
  #pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp][m], bb[cp][t], acc[m][t], 0, 0, 0);
      };
      // `touch` = an empty asm that reads a whole operand set: hipcc retires LDS reads with lgkmcnt(0) at the first use, so the
      // use is placed BEFORE the other set's fetch is issued -- each wait then only covers reads issued a full MFMA group earlier.
      auto touch = [&](const float (&a)[CC / 2][MT], const float (&bb)[CC / 2][NT]) {
  #pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp) {
  #pragma unroll
          for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(a[cp][m]));
  #pragma unroll
          for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(bb[cp][t]));
        }
        asm volatile("" ::: "memory");
      };
      load_ops(aA, bA);
      for (int slot = 0; slot + 1 < nv; slot += 2) {
        touch(aA, bA);
        load_ops(aB, bB);       // tap slot+1
        __builtin_amdgcn_sched_barrier(6);   // VALU / SALU may cross, LDS reads and MFMAs may not
        mfmas(aA, bA);          // tap slot
        __builtin_amdgcn_sched_barrier(0);
        touch(aB, bB);
        load_ops(aA, bA);       // tap slot+2 (wrapped, unused, when slot+2 == nv)
        __builtin_amdgcn_sched_barrier(6);
        mfmas(aB, bB);          // tap slot+1
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nv & 1) mfmas(aA, bA);
    }
    __builtin_amdgcn_sched_barrier(0);                           // keep the MFMAs of this chunk in front of the DMA wait
    __syncthreads();                                             // vmcnt(0): chunk+1 landed; barrier: this buffer is free
    if (single && chunk + 1 < p.nchunks) {
      issue(chunk + 1, 0);
      __syncthreads();
    }
  }

  g2_epilogue<MT, NT>(acc, p, bias, out, smem, tile_id, n, qd, q0h, q0w, wave, l31, hh, tid);
}



// ---------------------------------------------------------------------------------------------------------------------
// Exact-f32 products on the bf16 matrix pipe ("x9"), stride-1 convolutions (forward and transposed).
//
// Every fp32 operand is split EXACTLY into three bf16 values by truncation (x = hi + mid + lo: 8 + 8 + 8 significant bits); the
// partial products of a (weight, input) pair are exact in fp32 and are accumulated in fp32, smallest first, by one
// v_mfma_f32_32x32x16_bf16 each -- eight of the nine: lo x lo is below 2^-32 of the product and is dropped (X9_FIRST; 8 x 8 passes per 16
// reduction elements against 8 x 16 passes of v_mfma_f32_32x32x2_f32: half the matrix-pipe time).  The split has to be paid once per
// staged ELEMENT, not once per use (27 uses per element: that variant is VALU-bound), so the staging path differs
// from igemm2_kernel:
//   * a chunk is 4 input channels; the patch is fetched into REGISTERS (16-byte row segments of the 4 channels, prefetched one chunk
//     ahead, across the MFMA loop), split there (9 VALU per value pair) and written to LDS as [position][hi|mid|lo][4 channels]
//     bf16 -- 24 bytes per position, so ONE address serves the three components of a tap;
//   * the 16 reduction elements of an MFMA are 4 taps x 4 channels: a lane (column l31, K-half hh) reads the 8-byte channel
//     quadruplets of taps 4g + 2hh and 4g + 2hh + 1 for each component (positions from a per-tap offset table in LDS);
//   * the weights are split and laid out in fragment order by the pack kernel ([chunk][tap group][component][row tile][lane][8]) and
//     arrive by LDS-DMA while the patch is being split;
//   * the split of chunk i+1 is computed in the shadow of the last two tap groups' MFMAs of chunk i and kept in registers; between the
//     chunks only the 16-byte LDS stores remain (one patch buffer, two weight buffers, two resident workgroups per CU);
//   * 2-D kernels (9 taps) use 8-channel chunks and tap PAIRS (48 bytes per position, one ds_read_b128 per component): 10 instead
//     of 12 tap slots; layers dilated along H take the rows of one dilation phase per tile (G2P::rstep);
//   * NC = 1 (operand precision "bf16"): one component -- the operands rounded to bf16 (RNE) once per staged element.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// (x, y) -> packed bf16 pairs of the three components, x in the low half (conv_internal.h: round-to-nearest split)
__device__ __forceinline__ void split_pair(float x, float y, unsigned& h, unsigned& m, unsigned& l) { dpf_split_pair(x, y, h, m, l); }

// CC = 4: a group = 4 taps x 4 channels (3-D kernels: 27 taps -> 7 groups), two 256-unit rounds per chunk;
// CC = 8: a group = 2 taps x 8 channels (2-D kernels:  9 taps -> 5 groups), one round.
// first partial product of the nine (smallest first) that is issued: 3 (default, conv_internal.h) skips lo x lo, lo x mid, mid x lo
constexpr int X9_FIRST = DPF_X9_FIRST;
constexpr int X9_NP = 9 - X9_FIRST;

// NC = 3: the exact split (x9 / x8); NC = 1 (operand precision "bf16"): one component, the operand rounded to bf16 (RNE) -- the same
// staging and layouts with a third of the bytes and one MFMA per (row tile, position row, tap group).
// NC = 2 (dpf_set_f32_matrix_path(2)): two f16 components of the block-scaled operand (conv_internal.h), three MFMAs (lo*hi, hi*lo, hi*hi) on
// v_mfma_f32_32x32x16_f16.  The block is one channel chunk of the workgroup's patch: every lane takes the largest exponent of the values it
// fetched, the waves exchange theirs through LDS (one extra barrier per chunk, before the split), and the chunk is scaled by the exponent
// the accumulators already carry unless its own maximum lies above it or more than 3 bits below it (then the accumulators are rescaled,
// exactly).  RANGE GUARD: a lane holds all CC channels of its positions, so the scan that finds its largest exponent also yields the smallest
// non-zero per-position maximum; a position whose values all lie more than 2^17 below the scale is DEFERRED -- it contributes exact zeros
// to this pass and its values stay in registers -- and a chunk with a non-zero deferred position is contracted again (same weights), the
// deferred positions at their own scale: every position is contracted exactly once, in the pass whose scale lies within 2^17 of it, so
// every output position sees its inputs to fp32 precision relative to ITS OWN inputs, not to the tile's.  Chunks without such a position
// (all of them on ordinary data) pay 17 vector instructions per lane for the test; a wave that holds one splits its values a second time, masked.  (Deferred positions must be exact zeros, not small
// numbers: fed subnormal f16 values next to large operands the matrix core aligns its adder to the exponent FIELDS and drops accumulator
// bits -- tools/probes/mfma_f16_accum_probe.hip.)  The weights are scaled once per launch by the pack kernel; the epilogue multiplies by
// 2^-(both scales) exactly (v_ldexp_f32).
template <int CC, int NC> struct X9 {
  static constexpr int NU = CC == 4 ? 2 : 1;     // patch units (CC channels x 4 positions) per thread and chunk
  static constexpr int PB = 2 * NC * CC;         // bytes per position: [hi | mid | lo][CC] bf16
  static constexpr int TPG = 16 / CC;            // taps per group
  static constexpr int PD = PB / 4;              // dwords per position
};

// In-kernel s_memtime stamps (-DDPF_STAMPS builds only; tools/x9_stamps.py reads them back): per workgroup [start, first barrier,
// after the chunk loop, end, s_memrealtime at start, at end, XCC / CU id, chunks]
#ifdef DPF_STAMPS
__device__ unsigned long long g_x9_passes[4];      // [chunk passes, of them extra passes over deferred positions, accumulator rescales, tiles] of the f16-component launches
__device__ unsigned long long g_x9_stamps[8 * 16384];
#define X9_STAMP(slot, val) if (tid == 0 && blockIdx.x < 16384) g_x9_stamps[blockIdx.x * 8 + (slot)] = (val);
#define X9_COUNT(slot) if (tid == 0) atomicAdd(&g_x9_passes[slot], 1ull);
#else
#define X9_STAMP(slot, val)
#define X9_COUNT(slot)
#endif

template <int MT, int NT, int CC, bool SH, int NC = 3>
__global__ __launch_bounds__(256, 2) void igemm3_x9_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wpk,
                                                           const float* __restrict__ bias, float* __restrict__ out, G2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(NT == 2 || NT == 4, "position rows are processed in pairs");
  constexpr int KT = 32 * MT;
  constexpr int NU = X9<CC, NC>::NU, PB = X9<CC, NC>::PB, TPG = X9<CC, NC>::TPG, PD = X9<CC, NC>::PD;
  constexpr int NPROD = NC == 3 ? X9_NP : (NC == 2 ? 3 : 1);     // MFMAs per (row tile, position row, tap group)
  constexpr int SPV = NC == 3 ? 9 : (NC == 2 ? 6 : 1);            // vector instructions of the split per value pair
  constexpr int NPS = NT / 2;                    // pair steps (two position rows, interleaved accumulators) per group
  constexpr int NSL = NU * 4;                    // split slices (one position of one unit) per chunk; half of them per tail group
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int hh = lane >> 5;

  int b = (blockIdx.x & 7) * p.cpx + (blockIdx.x >> 3);           // XCD-aware tile order, as igemm2_kernel
  if (b >= p.ntiles) return;
  X9_STAMP(0, __builtin_readcyclecounter()) X9_STAMP(4, __builtin_amdgcn_s_memrealtime())
  const int tile_id = b;
  const int qd = (b % p.odt) * p.pz; b /= p.odt;
  const int tw = b % p.tilesW; b /= p.tilesW;
  const int th = b % p.tilesH;
  const int n = b / p.tilesH;
  // row-dilated tiles: tile row th = (block of thp * rstep rows, phase th % rstep); its rows are rstep apart
  const int q0h = (th / p.rstep) * (p.thp * p.rstep) + th % p.rstep, q0w = tw * 32;
  const int i0d = qd + p.e0d, i0h = q0h + p.e0h;
  const int a0 = q0w + p.e0w - p.colshift;

  const int TG = (p.T + TPG - 1) / TPG;
  const int patchBytes = PB * p.rpc * p.RS;
  const int wBytes = TG * NC * MT * 1024;
  char* s_patch = reinterpret_cast<char*>(smem);
  char* s_w = s_patch + patchBytes;                                // SH: two weight buffers
  int* s_tab = reinterpret_cast<int*>(s_w + (SH ? 2 : 1) * wBytes);
  int* s_red = s_tab + 32;                                         // NC = 2: [wave] largest exponent of the values in flight; [8 + parity] "a non-zero position was deferred"
  int Ex = DPF_H3_EMIN;                                            // NC = 2: the accumulators are in units of 2^(Ex + Ew - 282); Erun: largest chunk exponent of the tile so far
  int Erun = DPF_H3_EMIN;
  float scx = 0.f;                                                 //         multiplier of the values being split (chunk: 2^(141 - Enext); residual pass: 2^(Ex - Enext))
  unsigned tbC = 0, tbN = 0;                                       //         smallest position maximum (bit pattern, pv's units) the pass being contracted / prepared takes; 0: everything
  unsigned kminL = 0xffffffffu;                                    //         this lane's smallest non-zero position maximum of the values in flight (bit pattern - 1)
  const long long x_chan = (long long)p.ID * p.IH * p.IW;
  const float* xn = x + (long long)n * p.C * x_chan;
  const int nunits = p.rpc * p.SR;

  if (tid < 32) s_tab[tid] = tid < 28 ? PB * p.tapoff[tid] : 0;

  int goff[NU], loff[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const unsigned f = tid + 256 * j;
    const unsigned row = (f * p.mSR) >> 20;
    const int seg = f - row * p.SR;
    const unsigned pl = (row * p.mEH) >> 20;
    const int rr = row - pl * p.ext_h;
    const int id = i0d + (int)pl, ih = i0h + rr * p.rstep, iw = a0 + 4 * seg;
    const bool ok = (int)f < nunits && id >= 0 && id < p.ID && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
    goff[j] = ok ? (int)(((long long)id * p.IH + ih) * p.IW + iw) : -1;
    loff[j] = (int)f < nunits ? PB * ((int)row * p.RS + 4 * seg) : -1;
  }

  f32x4 pv[NU][CC];                  // raw patch values of the NEXT chunk (in flight across the MFMA loop)
  unsigned sp[NU][4][PD];            // their split form: [unit][position][hi: CC/2 dwords | mid | lo]
  auto prefetch = [&](int chunk) {
    const float* xc = xn + (long long)chunk * CC * x_chan;
    const int crem = p.C - chunk * CC;
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
      for (int ch = 0; ch < CC; ++ch) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (goff[j] >= 0 && ch < crem) v = *reinterpret_cast<const f32x4*>(xc + (long long)ch * x_chan + goff[j]);
        pv[j][ch] = v;
      }
  };
  auto issue_w = [&](int chunk, int buf) {
    const char* wc = reinterpret_cast<const char*>(wpk) + (long long)chunk * wBytes;
    char* dst = s_w + buf * wBytes;
    const int nws = wBytes >> 4;
    for (int f0 = wave * 64; f0 < nws; f0 += 256) {
      const int f = f0 + lane;
      if (f < nws) glds16(reinterpret_cast<const float*>(wc + f * 16), reinterpret_cast<float*>(dst + f0 * 16));
    }
  };
  auto split_slice = [&](int sl) {   // slice = (unit, position)
    const int j = sl >> 2, ps = sl & 3;
#pragma unroll
    for (int k = 0; k < CC / 2; ++k) {
      if constexpr (NC == 3) split_pair(pv[j][2 * k][ps], pv[j][2 * k + 1][ps], sp[j][ps][k], sp[j][ps][CC / 2 + k], sp[j][ps][CC + k]);
      else if constexpr (NC == 2) dpf_split_pair_h(pv[j][2 * k][ps] * scx, pv[j][2 * k + 1][ps] * scx, sp[j][ps][k], sp[j][ps][CC / 2 + k]);
      else sp[j][ps][k] = pk_bf16(pv[j][2 * k][ps], pv[j][2 * k + 1][ps]);
    }
  };
  // NC = 2: largest exponent of the values in flight (this lane's, then the wave's) -> s_red[wave]; after a barrier read_exp() is the
  // workgroup's.  The scan runs through the per-position maxima (a lane holds all CC channels of a position), whose smallest non-zero one
  // stays in kminL (bit pattern - 1: zero positions wrap to the top and drop out).
  auto pos_max = [&](int j, int ps) {
    float mp = __builtin_fabsf(pv[j][0][ps]);
#pragma unroll
    for (int ch = 1; ch < CC; ++ch) mp = __builtin_fmaxf(mp, __builtin_fabsf(pv[j][ch][ps]));
    return mp;
  };
  auto post_exp = [&]() {
    // per-position maxima, accumulated over channel PAIRS in the order the loads were issued (v_max3_f32: two channels per instruction), so
    // that only the last pair's four instructions and the folds below wait for the last load
    float pm[NU][4];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) pm[j][ps] = __builtin_fmaxf(__builtin_fabsf(pv[j][0][ps]), __builtin_fabsf(pv[j][1][ps]));
#pragma unroll
      for (int ch = 2; ch < CC; ch += 2)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) pm[j][ps] = __builtin_fmaxf(__builtin_fmaxf(pm[j][ps], __builtin_fabsf(pv[j][ch][ps])), __builtin_fabsf(pv[j][ch + 1][ps]));
    }
    float m = 0.f;
    kminL = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
      for (int ps = 0; ps < 4; ps += 2) {                           // two positions per step: v_max3_f32 / v_min3_u32 take both
        m = __builtin_fmaxf(__builtin_fmaxf(m, pm[j][ps]), pm[j][ps + 1]);
        kminL = min(min(kminL, __builtin_bit_cast(unsigned, pm[j][ps]) - 1u), __builtin_bit_cast(unsigned, pm[j][ps + 1]) - 1u);
      }
    const int e = dpf_wave_max_exp(__builtin_bit_cast(unsigned, m));
    if (lane == 0) s_red[wave] = e;
  };
  auto read_exp = [&]() {
    const int4 r = *reinterpret_cast<const int4*>(s_red);
    int e = max(max(r.x, r.y), max(r.z, r.w));
    e = __builtin_amdgcn_readfirstlane(e);
    return e > 254 ? 254 : (e < DPF_H3_EMIN ? DPF_H3_EMIN : e);    // (Inf / NaN inputs: the result is NaN either way)
  };
  auto tb_of = [&](int e) {                                         // bit pattern of 2^(e - 127), e clamped to the normal range
    return (unsigned)(e < 1 ? 1 : (e > 254 ? 254 : e)) << 23;
  };
  // a new chunk whose exponents were just exchanged: its scale (Enext), multiplier and range.  The accumulators' exponent is kept while
  // the chunk's maximum lies 0 .. 3 bits below it (no rescale); it never drops more than DPF_H3_MAXDROP below the tile's running maximum
  // (no overflow).
  int Enext = Ex;
  auto next_chunk_scale = [&]() {
    const int e = read_exp();
    Erun = e > Erun ? e : Erun;
    const int lo = Erun - DPF_H3_MAXDROP;
    Enext = (e > Ex || (e < Ex - 3 && e > DPF_H3_EMIN)) ? (e > lo ? e : lo) : Ex;      // (an all-zero chunk keeps the exponent)
    scx = dpf_h3_scale(Enext);
    tbN = p.guard ? tb_of(Enext - DPF_H3_RANGE) : 0u;
  };
  // another pass over the chunk just split: pv holds the deferred positions' values in units of the current scale (2^(141 - Ex)), zeros
  // elsewhere; their own scale on top of it.  The last pass takes whatever is left.
  auto next_residual_scale = [&](int pass_next) {
    const int e = read_exp();
    const int lo = Erun - DPF_H3_MAXDROP;
    const int en = Ex + e - 141;
    Enext = en > lo ? en : lo;
    if (Enext > Ex) Enext = Ex;
    scx = __builtin_bit_cast(float, (unsigned)(127 + Ex - Enext) << 23);
    tbN = pass_next < DPF_H3_MAXPASS ? tb_of(Enext - DPF_H3_RANGE + 141 - Ex) : 0u;
  };
  // does this lane hold a non-zero position below the range of the pass being prepared?  Posted to LDS for the workgroup ("another pass
  // follows": two alternating slots -- written while the pass is prepared, read after the barrier that publishes it, cleared at the top of
  // the following iteration); the wave-wide answer decides how the wave splits
  auto lane_defers = [&](int par) {
    const bool fl = tbN != 0u && kminL < tbN - 1u;
    if (fl) s_red[8 + par] = 1;
    return __builtin_amdgcn_ballot_w64(fl) != 0ull;
  };
  // the split of a wave with deferred positions (rare; overwrites the plain split): a deferred position contributes exact zeros
  auto split_masked = [&]() {
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const float ml = __builtin_bit_cast(unsigned, pos_max(j, ps)) >= tbN ? scx : 0.f;
#pragma unroll
        for (int k = 0; k < CC / 2; ++k) dpf_split_pair_h(pv[j][2 * k][ps] * ml, pv[j][2 * k + 1][ps] * ml, sp[j][ps][k], sp[j][ps][CC / 2 + k]);
      }
  };
  auto residual_update = [&]() {      // pv <- the positions the pass just published deferred, in units of its scale; zeros elsewhere
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const float ml = __builtin_bit_cast(unsigned, pos_max(j, ps)) >= tbC ? 0.f : scx;
#pragma unroll
        for (int ch = 0; ch < CC; ++ch) pv[j][ch][ps] *= ml;
      }
  };
  auto store_split = [&]() {
#pragma unroll
    for (int j = 0; j < NU; ++j)
      if (loff[j] >= 0) {
        u32x4* dst = reinterpret_cast<u32x4*>(s_patch + loff[j]);
#pragma unroll
        for (int i = 0; i < PD; ++i) {
          const int e = 4 * i;
          const u32x4 v = {sp[j][e / PD][e % PD], sp[j][(e + 1) / PD][(e + 1) % PD], sp[j][(e + 2) / PD][(e + 2) % PD], sp[j][(e + 3) / PD][(e + 3) % PD]};
          dst[i] = v;
        }
      }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][t][j] = 0.f;
  auto rescale_acc = [&](int de) {                                  // acc *= 2^de (exact)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[m][t][j] = __builtin_ldexpf(acc[m][t][j], de);
  };

  int lb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = wave * NT + t, rz = r >> p.thp_shift, ry = r & (p.thp - 1);
    lb[t] = PB * (rz * p.planeStride + ry * p.RS + l31 + p.colshift);
  }

  // SH (split in the shadow): the split of chunk i+1 is computed behind the last two tap groups' MFMAs of chunk i, its weights arrive in
  // the other weight buffer meanwhile, and only the LDS stores stand between two chunks.  !SH: one weight buffer; weights, split and
  // stores form a phase of their own, hidden by the co-resident workgroup's MFMAs.
  if constexpr (NC == 2) {
    if (tid < 2) s_red[8 + tid] = 0;
  }
  if constexpr (SH) issue_w(0, 0);
  prefetch(0);
  int par = 0;                                                     // parity of the pass being prepared (its slot of the deferral flag)
  if constexpr (SH) {
    bool wdef = false;
    if constexpr (NC == 2) {
      post_exp();
      __syncthreads();
      next_chunk_scale();
      Ex = Enext;
      wdef = lane_defers(par);
    }
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) split_slice(sl);
    if constexpr (NC == 2) {
      if (__builtin_expect(wdef, 0)) split_masked();
    }
  }

  // one iteration = one pass over a chunk; NC = 2: a chunk with deferred positions takes further passes (pass > 0) before the next chunk
  // is fetched -- the deferred values stay in pv
  int pass = 0;
  for (int chunk = 0; chunk < p.nchunks;) {
    if constexpr (!SH) {
      bool wdef = false;
      if (pass == 0) issue_w(chunk, 0);
      if constexpr (NC == 2) {
        post_exp();
        __syncthreads();
        if (pass == 0) next_chunk_scale(); else next_residual_scale(pass);
        wdef = lane_defers(par);
      }
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) split_slice(sl);
      if constexpr (NC == 2) {
        if (__builtin_expect(wdef, 0)) split_masked();
      }
    }
    bool more = false;                                             // (workgroup-uniform) another pass over this chunk follows
    int dflag = 0;
    if constexpr (NC == 2) {
      if constexpr (SH) {
        if (chunk > 0 || pass > 0) dflag = s_red[8 + par];          // (posted before the barrier that ended the previous iteration; the first pass of a tile: below)
      }
      if (tid == 0) s_red[8 + (par ^ 1)] = 0;
      if (__builtin_expect(Enext != Ex, 0)) { rescale_acc(Ex - Enext); Ex = Enext; X9_COUNT(2) }    // the pass about to be contracted changes the accumulators' exponent
      X9_COUNT(0)
      if (pass > 0) { X9_COUNT(1) }
      tbC = tbN;
    }
    store_split();
    __syncthreads();                                               // vmcnt(0): this chunk's weights landed; barrier: patch written
    if (chunk == 0) { X9_STAMP(1, __builtin_readcyclecounter()) }
    const char* s_wc = s_w + (SH ? (chunk & 1) * wBytes : 0);
    if constexpr (NC == 2) {
      if (!SH || (chunk == 0 && pass == 0)) dflag = s_red[8 + par]; // (posted before the barrier just passed)
      more = pass < DPF_H3_MAXPASS && __builtin_amdgcn_readfirstlane(dflag) != 0;
      par ^= 1;
    }
    if (__builtin_expect(more, 0)) {
      residual_update();
    } else if (chunk + 1 < p.nchunks) {
      if constexpr (SH) issue_w(chunk + 1, (chunk + 1) & 1);
      prefetch(chunk + 1);
    }

    u32x4 aC[NC][MT], aN[NC][MT], bP[2][2][NC];                       // bP[set][row of the pair][component]
    int oC0, oC1 = 0, oN0 = 0, oN1 = 0;
    auto load_a = [&](int g, u32x4 (&a)[NC][MT]) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) a[c][m] = *reinterpret_cast<const u32x4*>(s_wc + ((g * NC + c) * MT + m) * 1024 + lane * 16);
    };
    auto load_t = [&](int g, int& o0, int& o1) {
      if constexpr (CC == 4) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(s_tab + 4 * g + 2 * hh);
        o0 = (int)v[0]; o1 = (int)v[1];
      } else {
        o0 = s_tab[2 * g + hh];
      }
    };
    auto load_b = [&](int t, int o0, int o1, u32x4 (&bb)[NC]) {
      if constexpr (CC == 4 && NC == 2) {                           // 16 bytes per position: [hi x 4 | lo x 4] of a tap in one read
        const u32x4 q0 = *reinterpret_cast<const u32x4*>(s_patch + lb[t] + o0), q1 = *reinterpret_cast<const u32x4*>(s_patch + lb[t] + o1);
        const u32x4 vh = {q0[0], q0[1], q1[0], q1[1]}, vl = {q0[2], q0[3], q1[2], q1[3]};
        bb[0] = vh; bb[1] = vl;
      } else if constexpr (CC == 4) {
        const char* p0 = s_patch + lb[t] + o0;
        const char* p1 = s_patch + lb[t] + o1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const u32x2 q0 = *reinterpret_cast<const u32x2*>(p0 + 8 * c), q1 = *reinterpret_cast<const u32x2*>(p1 + 8 * c);
          const u32x4 v = {q0[0], q0[1], q1[0], q1[1]};
          bb[c] = v;
        }
      } else {
        const char* p0 = s_patch + lb[t] + o0;
#pragma unroll
        for (int c = 0; c < NC; ++c) bb[c] = *reinterpret_cast<const u32x4*>(p0 + 16 * c);
      }
    };
    auto touch_pair = [&](const u32x4 (&bb)[2][NC]) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" ::"v"(bb[r][c]));
      asm volatile("" ::: "memory");
    };
    auto touch_a = [&](const u32x4 (&a)[NC][MT]) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(a[c][m]));
    };
    auto mfma_pair = [&](const u32x4 (&a)[NC][MT], const u32x4 (&bb)[2][NC], int t0) {
      // component index 0 = hi, 1 = mid, 2 = lo; smallest partial products first; the two position rows (and the row tiles) alternate,
      // so consecutive MFMAs never accumulate into the same registers
      constexpr int oa[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
      constexpr int ob[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
      if constexpr (NC == 2) {
        constexpr int ha[3] = {1, 0, 0}, hb[3] = {0, 1, 0};         // lo*hi, hi*lo, hi*hi
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              acc[m][t0 + r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ha[i]][m]), __builtin_bit_cast(f16x8, bb[r][hb[i]]),
                                                                       acc[m][t0 + r], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = (NC == 3 ? X9_FIRST : 8); i < 9; ++i)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              acc[m][t0 + r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[NC == 3 ? oa[i] : 0][m]),
                                                                        __builtin_bit_cast(bf16x8, bb[r][NC == 3 ? ob[i] : 0]), acc[m][t0 + r], 0, 0, 0);
      }
    };
    // one tap group; HALF = 0 / 1: the first / second half of the next chunk's split slices is computed in the MFMAs' shadow
    auto group = [&](int g, auto half_c) {
      constexpr int HALF = decltype(half_c)::value;
      const int gn = g + 1 < TG ? g + 1 : 0;                      // the wrapped fetch of the last group is valid and unused
      load_t(gn, oN0, oN1);
#pragma unroll
      for (int s = 0; s < NPS; ++s) {
        touch_pair(bP[s & 1]);
        if (s == 0) touch_a(aC);
        if (s + 1 < NPS) {
          load_b(2 * s + 2, oC0, oC1, bP[(s + 1) & 1][0]);
          load_b(2 * s + 3, oC0, oC1, bP[(s + 1) & 1][1]);
        } else {
          load_a(gn, aN);
          load_b(0, oN0, oN1, bP[(s + 1) & 1][0]);
          load_b(1, oN0, oN1, bP[(s + 1) & 1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HALF >= 0) {
          constexpr int per = NSL / 2 / NPS;                       // slices per pair step
#pragma unroll
          for (int q = 0; q < per; ++q) split_slice(HALF * (NSL / 2) + s * per + q);
        }
        mfma_pair(aC, bP[s & 1], 2 * s);
        if constexpr (HALF >= 0) {
          constexpr int NM = 2 * NPROD * MT;
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (NSL / 2 / NPS * CC / 2 * SPV + NM - 1) / NM + 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) aC[c][m] = aN[c][m];
      oC0 = oN0; oC1 = oN1;
      if constexpr (NPS & 1) {                                     // an odd number of pair steps leaves the next group's rows in set 1
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < NC; ++c) bP[0][r][c] = bP[1][r][c];
      }
    };
    load_a(0, aC);
    load_t(0, oC0, oC1);
    load_b(0, oC0, oC1, bP[0][0]);
    load_b(1, oC0, oC1, bP[0][1]);
    if constexpr (SH) {
      for (int g = 0; g + 2 < TG; ++g) group(g, std::integral_constant<int, -1>{});
      bool wdef = false;
      if constexpr (NC == 2) {
        if (more || chunk + 1 < p.nchunks) {                       // the next pass's values (what is left of this chunk / the next chunk, landed by now): agree on their scale
          post_exp();
          __syncthreads();
          if (__builtin_expect(more, 0)) next_residual_scale(pass + 1); else next_chunk_scale();
          wdef = lane_defers(par);
        }
      }
      group(TG - 2, std::integral_constant<int, 0>{});
      group(TG - 1, std::integral_constant<int, 1>{});
      if constexpr (NC == 2) {
        if (__builtin_expect(wdef, 0)) split_masked();                                   // (wave-uniform, rare) this wave holds deferred positions: split again, masked
      }
    } else {
      for (int g = 0; g < TG; ++g) group(g, std::integral_constant<int, -1>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                               // the patch buffer is free
    if (more) ++pass;
    else { ++chunk; pass = 0; }
  }
  X9_STAMP(2, __builtin_readcyclecounter())
  if constexpr (NC == 2) { X9_COUNT(3) }
  if constexpr (NC == 2) {                                         // back to the operands' units (exact): the tile's exponent + the one of each OUTPUT ROW's weights
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int4 we = *reinterpret_cast<const int4*>(p.wexp + 32 * m + 8 * q4 + 4 * hh);       // accumulator register j <-> row (j & 3) + 8 (j >> 2) + 4 hh
        const int w4[4] = {we.x, we.y, we.z, we.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[m][t][4 * q4 + r] = __builtin_ldexpf(acc[m][t][4 * q4 + r], Ex + w4[r] - 282);
      }
  }
  g2_epilogue<MT, NT>(acc, p, bias, out, smem, tile_id, n, qd, q0h, q0w, wave, l31, hh, tid);
  X9_STAMP(3, __builtin_readcyclecounter()) X9_STAMP(5, __builtin_amdgcn_s_memrealtime()) X9_STAMP(6, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_ID, all 32 bits */)) X9_STAMP(7, (unsigned long long)p.nchunks)
}

// f16 components (NC = 2): one workgroup per OUTPUT ROW k of the launch (an output channel of a forward conv, an input channel of a data
// gradient).  The row's largest exponent -> wexp[k] (the conv's epilogue undoes it per accumulator row), then the row's weights times
// 2^(141 - that exponent) as (hi, lo) f16 components into the fragment layout below.  A scale per row instead of per tensor: a row of small
// weights keeps its full precision next to a row of large ones (the range guard of the weight operand), and the separate exponent kernel
// of round 5 is gone.
__global__ __launch_bounds__(256) void igemm3_pack_x9h_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int wA, int wB, int T, int TG, int MT,
                                                              int CC, int nchunks, int mode, int k0, int K, int C, int* __restrict__ wexp) {
  __shared__ int s_e[4];
  const int k = blockIdx.x, m = k >> 5, l31 = k & 31;
  const bool rowok = k < K;
  auto wat = [&](int c, int u) {
    const int a = mode == 0 ? k0 + k : c, bb = mode == 0 ? c : k0 + k;
    return w[((long long)a * wB + bb) * T + u];
  };
  float mx = 0.f;
  if (rowok)
    for (int i = threadIdx.x; i < C * T; i += 256) mx = __builtin_fmaxf(mx, __builtin_fabsf(wat(i / T, i % T)));
  const int e = dpf_wave_max_exp(__builtin_bit_cast(unsigned, mx));
  if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6] = e;
  __syncthreads();
  int E = max(max(s_e[0], s_e[1]), max(s_e[2], s_e[3]));
  E = E < DPF_H3_EMIN ? DPF_H3_EMIN : (E > 254 ? 254 : E);
  if (threadIdx.x == 0) wexp[k] = E;
  const float wscale = dpf_h3_scale(E);
  const int TPG = 16 / CC;
  for (int e2 = threadIdx.x; e2 < nchunks * TG * 16; e2 += 256) {
    const int i = e2 & 7, hh = (e2 >> 3) & 1, g = (e2 >> 4) % TG, chunk = (e2 >> 4) / TG;
    const int u = CC == 4 ? 4 * g + 2 * hh + (i >> 2) : 2 * g + hh;
    const int c = CC == 4 ? chunk * 4 + (i & 3) : chunk * 8 + i;
    const float v = (rowok && u < T && c < C) ? wat(c, u) * wscale : 0.f;
    unsigned sh, sl;
    dpf_split_pair_h(v, 0.f, sh, sl);
    const long long base = (((long long)(chunk * TG + g) * 2) * MT + m) * 512 + (hh * 32 + l31) * 8 + i;
    wpk[base] = (unsigned short)(sh & 0xffffu);
    wpk[base + 512LL * MT] = (unsigned short)(sl & 0xffffu);
  }
  (void)TPG;
}

// x9 weights: shorts [chunk][tap group g][component][row tile m][lane][8]; value i of lane (l31, hh) = component of
// w(out = k0 + 32 m + l31, reduce, tap), zero beyond T / C / K, with
//   CC = 4: reduce = 4 chunk + (i & 3), tap = 4 g + 2 hh + (i >> 2);      CC = 8: reduce = 8 chunk + i, tap = 2 g + hh
__global__ void igemm3_pack_x9_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int wA, int wB, int T, int TG, int MT, int CC,
                                      int NC, int nchunks, int mode, int k0, int K, int C) {
  const long long total = (long long)nchunks * TG * MT * 512;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e & 7), ln = (int)((e >> 3) & 63);
    const int m = (int)((e >> 9) % MT);
    const int g = (int)((e / (512LL * MT)) % TG);
    const int chunk = (int)(e / (512LL * MT * TG));
    const int u = CC == 4 ? 4 * g + 2 * (ln >> 5) + (i >> 2) : 2 * g + (ln >> 5);
    const int c = CC == 4 ? chunk * 4 + (i & 3) : chunk * 8 + i;
    const int k = m * 32 + (ln & 31);
    float v = 0.f;
    if (u < T && c < C && k < K) {
      const int a = mode == 0 ? k0 + k : c;
      const int bb = mode == 0 ? c : k0 + k;
      v = w[((long long)a * wB + bb) * T + u];
    }
    unsigned sh, sm, sl;
    dpf_split_pair(v, 0.f, sh, sm, sl);
    const long long base = (((long long)(chunk * TG + g) * NC) * MT + m) * 512 + ln * 8 + i;
    if (NC == 1) {                                                 // operand precision "bf16": round to nearest even
      wpk[base] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
    } else {
      wpk[base] = (unsigned short)(sh & 0xffffu);
      wpk[base + 512LL * MT] = (unsigned short)(sm & 0xffffu);
      wpk[base + 1024LL * MT] = (unsigned short)(sl & 0xffffu);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 transposed 3x3x3 convolution (pad 1, dilation 1): data gradient of the hourglass' stride-2 convs and the forward of
// its nn.ConvTranspose3d layers (src/model/stereodpnet/modules.py:215-227).  out[n,k,2q+r] = sum_{c,t} W'[c,k,t] x[n,c,q+e(r,t)]:
// per dimension the output parity r = 0 meets one tap (t = 1, e = 0) and r = 1 meets two (t = 0, e = 1; t = 2, e = 0).  The
// first-generation kernel ran the 8 parity classes as 8 workgroups that each re-staged the same input patch (36.7 TFLOP/s);
// here ONE workgroup stages the patch of a 4 x 32 q-tile once per channel chunk and keeps the accumulators of all 8 classes
// (8 x 16 registers per wave, one q-row per wave): 27 (class, tap) MFMAs per channel pair, their B operands being just the 8
// shifted views e in {0,1}^3 of the patch.  Same LDS-DMA double buffering and zero-page padding as igemm2_kernel.
struct T2P {
  int N, C, K, Ktot, k0;
  int ID, IH, IW, OD, OH, OW;
  int RS, SR, rpc, chanStride, nseg, nwseg, nchunks;
  int tilesH, tilesW, ntiles, cpx, QD;
  unsigned mSR, mRPC, mEH;
  int vec;                      // 16-byte stores allowed (DPF_G2_VEC_STORE)
};

struct T2Combo { int cls, tap; };
struct T2Unit { int e, n; T2Combo c[8]; };
struct T2Table { T2Unit u[8]; };

// units in issue order (sizes 8,4,4,2,2,2,1,4): shift e = (ed, eh, ew); per dimension shift 0 serves (r=0, t=1) and (r=1, t=2),
// shift 1 serves (r=1, t=0)
constexpr T2Table t2_table() {
  T2Table tb{};
  const int order[8] = {0, 1, 2, 3, 5, 6, 7, 4};
  for (int ui = 0; ui < 8; ++ui) {
    const int e = order[ui];
    const int es[3] = {e >> 2, (e >> 1) & 1, e & 1};
    T2Unit& u = tb.u[ui];
    u.e = e;
    u.n = 0;
    for (int a = 0; a < 2; ++a) {
      if (es[0] == 1 && a == 1) continue;
      for (int b = 0; b < 2; ++b) {
        if (es[1] == 1 && b == 1) continue;
        for (int c = 0; c < 2; ++c) {
          if (es[2] == 1 && c == 1) continue;
          // option index 0: (shift 0 -> r=0,t=1 | shift 1 -> r=1,t=0); option 1 (shift 0 only): r=1,t=2
          const int rd = es[0] == 1 ? 1 : a, td = es[0] == 1 ? 0 : (a ? 2 : 1);
          const int rh = es[1] == 1 ? 1 : b, th = es[1] == 1 ? 0 : (b ? 2 : 1);
          const int rw = es[2] == 1 ? 1 : c, tw = es[2] == 1 ? 0 : (c ? 2 : 1);
          u.c[u.n].cls = rd * 4 + rh * 2 + rw;
          u.c[u.n].tap = (td * 3 + th) * 3 + tw;
          ++u.n;
        }
      }
    }
  }
  return tb;
}

template <int CC, bool BF = false>
__global__ __launch_bounds__(256, 2) void igemm2_tr2_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                            const float* __restrict__ bias, float* __restrict__ out, T2P p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KT = 32;
  constexpr T2Table TB = t2_table();
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = (blockIdx.x & 7) * p.cpx + (blockIdx.x >> 3);
  if (b >= p.ntiles) return;
  const int qd = b % p.QD; b /= p.QD;
  const int tw = b % p.tilesW; b /= p.tilesW;
  const int th = b % p.tilesH;
  const int n = b / p.tilesH;
  const int q0h = th * 4, q0w = tw * 32;

  static_assert(!BF || CC == 8, "the bf16 path contracts 8 channels per MFMA");
  const int patchFloats = CC * p.chanStride;
  const int bufFloats = patchFloats + 4 * p.nwseg;
  const long long x_chan = (long long)p.ID * p.IH * p.IW;
  const float* xn = x + (long long)n * p.C * x_chan;

  int goff[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const unsigned f = tid + 256 * j;
    const unsigned row = (f * p.mSR) >> 20;
    const int seg = f - row * p.SR;
    const unsigned cc = (row * p.mRPC) >> 20;
    const unsigned rem = row - cc * p.rpc;
    const unsigned pl = (rem * p.mEH) >> 20;            // ext_h = 5 rows per plane
    const int rr = rem - pl * 5;
    const int id = qd + (int)pl, ih = q0h + rr, iw = q0w + 4 * seg;
    const bool ok = (int)f < p.nseg && id < p.ID && ih < p.IH && iw < p.IW;
    goff[j] = ok ? (int)((long long)cc * x_chan + ((long long)id * p.IH + ih) * p.IW + iw) : -1;
  }
  auto issue = [&](int chunk, int buf) {
    float* dbase = smem + buf * bufFloats;
    const float* xc = xn + (long long)chunk * CC * x_chan;
    const int crem = p.C - chunk * CC;
    const int flimit = (crem < CC ? crem : CC) * p.rpc * p.SR;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      if (j * 256 < p.nseg) {
        const int f = tid + 256 * j;
        if (f < p.nseg) {
          const float* src = (goff[j] >= 0 && f < flimit) ? xc + goff[j] : wpk;
          glds16(src, dbase + (j * 256 + wave * 64) * 4);
        }
      }
    }
    const float* wc = wpk + ZPAGE + (long long)chunk * p.nwseg * 4;
    float* wbase = dbase + patchFloats;
    for (int f0 = wave * 64; f0 < p.nwseg; f0 += 256) {
      const int f = f0 + lane;
      if (f < p.nwseg) glds16(wc + f * 4, wbase + f0 * 4);
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;

  // B operand of shift e for this lane: patch[(2cp+hh)][ed][wave + eh][l31 + ew]
  const int planeStride = 5 * p.RS;
  int shoff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) shoff[e] = (e >> 2) * planeStride + (wave + ((e >> 1) & 1)) * p.RS + l31 + (e & 1) + hh * (BF ? 4 : 1) * p.chanStride;
  const int abase = hh * KT + l31;
  const int chanStride = p.chanStride;

  issue(0, 0);
  __syncthreads();
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < p.nchunks) issue(chunk + 1, buf ^ 1);
    const float* s_in = smem + buf * bufFloats;
    if constexpr (BF) {
      // bf16 operands: a unit is one shift e; its B operand is the lane's 4 channels at that shift (rounded to bf16), its A operands
      // the packed bf16 weights [tap][half][k][4] of the (class, tap) combinations the shift serves: one 32x32x8 MFMA each
      const short* s_wb = reinterpret_cast<const short*>(s_in + patchFloats) + (hh * KT + l31) * 4;
      s16x4 aA[8], aB[8];
      float bA[4], bB[4];
      auto load_unit = [&](int ui, s16x4 (&a)[8], float (&bv)[4]) {
        const T2Unit& u = TB.u[ui];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = s_in[shoff[u.e] + i * chanStride];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < u.n) a[i] = *reinterpret_cast<const s16x4*>(s_wb + u.c[i].tap * (8 * KT));
      };
      auto mfma_unit = [&](int ui, const s16x4 (&a)[8], const float (&bv)[4]) {
        const T2Unit& u = TB.u[ui];
        const u32x2 pk = {pk_bf16(bv[0], bv[1]), pk_bf16(bv[2], bv[3])};
        const s16x4 b = __builtin_bit_cast(s16x4, pk);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < u.n) acc[u.c[i].cls] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a[i], b, acc[u.c[i].cls], 0, 0, 0);
      };
      auto touch = [&](int ui, const s16x4 (&a)[8], const float (&bv)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(bv[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < TB.u[ui].n) asm volatile("" ::"v"(a[i]));
        asm volatile("" ::: "memory");
      };
      load_unit(0, aA, bA);
#pragma unroll
      for (int ui = 0; ui < 8; ++ui) {
        if ((ui & 1) == 0) {
          touch(ui, aA, bA);
          if (ui + 1 < 8) load_unit(ui + 1, aB, bB);
          __builtin_amdgcn_sched_barrier(6);
          mfma_unit(ui, aA, bA);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          touch(ui, aB, bB);
          if (ui + 1 < 8) load_unit(ui + 1, aA, bA);
          __builtin_amdgcn_sched_barrier(6);
          mfma_unit(ui, aB, bB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      const float* s_w = s_in + patchFloats + abase;
      // units of (channel pair, shift) software pipelined over two register sets
      float aA[8], aB[8], bA, bB;
      auto load_unit = [&](int cp, int ui, float (&a)[8], float& bv) {
        const T2Unit& u = TB.u[ui];
        bv = s_in[shoff[u.e] + (2 * cp) * chanStride];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < u.n) a[i] = s_w[(u.c[i].tap * CC + 2 * cp) * KT];
      };
      auto mfma_unit = [&](int ui, const float (&a)[8], float bv) {
        const T2Unit& u = TB.u[ui];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < u.n) acc[u.c[i].cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv, acc[u.c[i].cls], 0, 0, 0);
      };
      auto touch = [&](int ui, const float (&a)[8], float bv) {
        asm volatile("" ::"v"(bv));
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < TB.u[ui].n) asm volatile("" ::"v"(a[i]));
        asm volatile("" ::: "memory");
      };
      load_unit(0, 0, aA, bA);
#pragma unroll
      for (int cp = 0; cp < CC / 2; ++cp) {
#pragma unroll
        for (int ui = 0; ui < 8; ++ui) {
          const int ncp = ui + 1 < 8 ? cp : cp + 1, nui = ui + 1 < 8 ? ui + 1 : 0;
          if ((ui & 1) == 0) {
            touch(ui, aA, bA);
            if (ncp < CC / 2) load_unit(ncp, nui, aB, bB);
            __builtin_amdgcn_sched_barrier(6);
            mfma_unit(ui, aA, bA);
            __builtin_amdgcn_sched_barrier(0);
          } else {
            touch(ui, aB, bB);
            if (ncp < CC / 2) load_unit(ncp, nui, aA, bA);
            __builtin_amdgcn_sched_barrier(6);
            mfma_unit(ui, aB, bB);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: classes (rd, rh, 0) and (rd, rh, 1) interleave along W: one 8-byte store per lane covers ow = 2q, 2q+1
  const int qh = q0h + wave, qw = q0w + l31;
  const long long out_plane = (long long)p.OH * p.OW;
  const long long kstride = (long long)p.OD * out_plane;
  float* on = out + ((long long)n * p.Ktot + p.k0) * kstride;
  const bool pair_ok = ((p.OW & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  const bool quad_ok = ((p.OW & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && p.vec;
#pragma unroll
  for (int rd = 0; rd < 2; ++rd)
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      const int od = 2 * qd + rd, oh = 2 * qh + rh, ow = 2 * qw;
      if (od >= p.OD || oh >= p.OH || ow >= p.OW) continue;
      float* op = on + ((long long)od * p.OH + oh) * p.OW + ow;
      if (quad_ok) {
        // 16-byte stores: lanes 2c and 2c + 1 hold ow = 4c .. 4c + 3 of rows j and j + 1; one DPP exchange hands the even lane row j's
        // four values and the odd lane row j + 1's (the store tail of narrow stores: g2_epilogue)
        const bool odd = (l31 & 1) != 0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const float a0 = acc[rd * 4 + rh * 2][j], a1 = acc[rd * 4 + rh * 2 + 1][j];
          const float b0 = acc[rd * 4 + rh * 2][j + 1], b1 = acc[rd * 4 + rh * 2 + 1][j + 1];
          const float s0 = odd ? a0 : b0, s1 = odd ? a1 : b1;                 // what the partner needs
          const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xf, 0xf, true));
          const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xf, 0xf, true));
          const int k = ((j + (odd ? 1 : 0)) & 3) + 8 * (j >> 2) + 4 * hh;
          if (k < p.K) {
            const float bv = bias ? bias[p.k0 + k] : 0.f;
            const f32x4 v = odd ? f32x4{t0 + bv, t1 + bv, b0 + bv, b1 + bv} : f32x4{a0 + bv, a1 + bv, t0 + bv, t1 + bv};
            *reinterpret_cast<f32x4*>(on + ((long long)od * p.OH + oh) * p.OW + (ow & ~3) + (long long)k * kstride) = v;
          }
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = (j & 3) + 8 * (j >> 2) + 4 * hh;
        if (k < p.K) {
          const float bv = bias ? bias[p.k0 + k] : 0.f;
          const float v0 = acc[rd * 4 + rh * 2][j] + bv, v1 = acc[rd * 4 + rh * 2 + 1][j] + bv;
          float* o = op + (long long)k * kstride;
          if (pair_ok) {
            *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
          } else {
            o[0] = v0;
            if (ow + 1 < p.OW) o[1] = v1;
          }
        }
      }
    }
}

// wpk[0, ZPAGE) = 0;  wpk[ZPAGE + ((chunk*T + t)*CC + cc)*KT + k] = w(out = k0 + k, reduce = chunk*CC + cc, tap = t), zero padded
__global__ void igemm2_pack_kernel(const float* __restrict__ w, float* __restrict__ wpk, int wA, int wB, int T, int KT, int CC, int nchunks,
                                   int mode, int k0, int K, int C) {
  const long long total = (long long)nchunks * T * CC * KT + ZPAGE;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i >= ZPAGE) {
      const long long e = i - ZPAGE;
      const int k = (int)(e % KT);
      const int cc = (int)((e / KT) % CC);
      const int t = (int)((e / ((long long)KT * CC)) % T);
      const int chunk = (int)(e / ((long long)KT * CC * T));
      const int c = chunk * CC + cc;
      if (k < K && c < C) {
        const int a = mode == 0 ? k0 + k : c;
        const int bb = mode == 0 ? c : k0 + k;
        v = w[((long long)a * wB + bb) * T + t];
      }
    }
    wpk[i] = v;
  }
}

// bf16 operand path: wpk[0, ZPAGE floats) = 0; then shorts [chunk][tap][half][k][4] = bf16(w(out = k0 + k, reduce = 8*chunk + 4*half + i, tap))
__global__ void igemm2_pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int wA, int wB, int T, int KT, int nchunks,
                                        int mode, int k0, int K, int C) {
  const long long total = (long long)nchunks * T * 8 * KT + 2 * ZPAGE;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i >= 2 * ZPAGE) {
      const long long e = i - 2 * ZPAGE;
      const int ii = (int)(e & 3);
      const int k = (int)((e >> 2) % KT);
      const int half = (int)((e / (4LL * KT)) & 1);
      const int t = (int)((e / (8LL * KT)) % T);
      const int chunk = (int)(e / (8LL * KT * T));
      const int c = chunk * 8 + half * 4 + ii;
      if (k < K && c < C) {
        const int a = mode == 0 ? k0 + k : c;
        const int bb = mode == 0 ? c : k0 + k;
        v = w[((long long)a * wB + bb) * T + t];
      }
    }
    wpk[i] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
  }
}

unsigned magic20(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

template <int MT, int NT, int CC, bool BF = false>
int launch_g2(const float* x, const float* wpk, const float* bias, float* out, const G2P& p, size_t lds, long long blocks, hipStream_t st) {
  if (lds > 48 * 1024) {
    static bool done = false;   // per instantiation
    if (!done) {
      if (hipFuncSetAttribute((const void*)igemm2_kernel<MT, NT, CC, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return DPF_ERR_LAUNCH;
      done = true;
    }
  }
  hipLaunchKernelGGL((igemm2_kernel<MT, NT, CC, BF>), dim3((unsigned)blocks), dim3(256), lds, st, x, wpk, bias, out, p);
  return dpf_check_launch();
}


template <int MT, int NT, int CC, bool SH, int NC>
int launch_x9(const float* x, const unsigned short* wpk, const float* bias, float* out, const G2P& p, size_t lds, long long blocks, hipStream_t st) {
  static bool done = false;   // per instantiation
  if (!done) {
    if (hipFuncSetAttribute((const void*)igemm3_x9_kernel<MT, NT, CC, SH, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DPF_ERR_LAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((igemm3_x9_kernel<MT, NT, CC, SH, NC>), dim3((unsigned)blocks), dim3(256), lds, st, x, wpk, bias, out, p);
  return dpf_check_launch();
}

// One launch of igemm3_x9_kernel for the output channels [k_off, k_off + kn) of the launch `d` (kn <= 64); `p` carries the
// tile-independent geometry (strides, e0*, RS, colshift).  DPF_ERR_UNSUPPORTED: not eligible, nothing was launched.
// mt_fit: row tiles the tile geometry (depth split, weight buffers) is chosen for -- the slices of one K > 64 launch pass the LARGEST slice's
// count, so that every slice picks the same depth split / tile order and the shared BatchNorm statistics slab has one row numbering.
int x9_try(const float* x, const float* w, const float* bias, float* out, float* ws, const DpfConvDesc& d, const G2P& p, int k_off, int kn, int NT,
           DpfConvStats* stats, hipStream_t st, int NC, int mt_fit = 0) {
  static const int x9_on = env_int("DPF_IGEMM3", 1), x9_min_c = 8, x9_cc = env_int("DPF_IGEMM3_CC", 0),
                   x9_sh = env_int("DPF_IGEMM3_SH", -1);
  const int T = d.kd * d.kh * d.kw, MT = (kn + 31) / 32, TH = 4 * NT;
  if (!x9_on || (NC >= 2 && !dpf_conv_f32_x9()) || MT > 2 || NT * MT > 4 || d.C < x9_min_c) return DPF_ERR_UNSUPPORTED;
  G2P q = p;
  q.K = kn; q.k0 = d.k0 + k_off;
  // 2-D layers dilated along H: a tile takes the rows of ONE dilation phase (dh apart), so its patch is thp + kh - 1 image rows fetched dh
  // apart instead of thp + (kh - 1) * dh consecutive ones (dilation 8: 18 rows instead of 32 -- within the staging budget)
  static const int rstep_on = env_int("DPF_IGEMM3_RSTEP", 1);
  q.rstep = (rstep_on && d.kd == 1 && d.kh > 1 && d.dh > 1) ? d.dh : 1;
  const int dhl = d.dh / q.rstep;
  // chunk layout with the fewest tap slots (4 channels x tap quadruples or 8 channels x tap pairs); ties: the smaller patch
  int CC9 = ((T + 1) / 2) * 2 < ((T + 3) / 4) * 4 ? 8 : 4;
  if (x9_cc == 4 || x9_cc == 8) CC9 = x9_cc;
  const int NU9 = CC9 == 4 ? 2 : 1, PB9 = 2 * NC * CC9;
  // depth split: among the splits whose patch fits the per-thread unit budget, the one with the fewest staged rows
  auto set_pz = [&](int pz) {
    q.pz = pz; q.thp = TH / pz; q.thp_shift = 0;
    while ((1 << q.thp_shift) < q.thp) ++q.thp_shift;
    q.odt = dpf_div_up(d.OD, pz);
    q.ext_d = (pz - 1) + (d.kd - 1) * d.dd + 1;
    q.ext_h = (q.thp - 1) + (d.kh - 1) * dhl + 1;
    q.planeStride = q.ext_h * q.RS; q.SR = q.RS / 4; q.rpc = q.ext_d * q.ext_h;
    return q.rpc * q.SR;
  };
  const int TG = (T + 16 / CC9 - 1) / (16 / CC9);
  int best = 0, best_units = 1 << 30, sh = 0;
  for (int shc : {1, 0}) {                                     // split in the MFMAs' shadow (two weight buffers) when the LDS has room
    if (best || (x9_sh >= 0 && shc != x9_sh)) continue;
    for (int pz : {1, 2, 4}) {
      if (!(pz == 1 || (d.kd > 1 && pz <= TH / 2 && pz <= d.OD))) continue;
      const int units = set_pz(pz);
      const size_t l9 = (size_t)PB9 * q.rpc * q.RS + (size_t)(shc ? 2 : 1) * TG * NC * (mt_fit > MT ? mt_fit : MT) * 1024 + 256;
      if (units <= NU9 * 256 && 2 * l9 <= 160 * 1024 && units < best_units) { best = pz; best_units = units; sh = shc; }
    }
  }
  if (!best || TG < 2) return DPF_ERR_UNSUPPORTED;
  set_pz(best);
  const size_t lds9 = (size_t)PB9 * q.rpc * q.RS + (size_t)(sh ? 2 : 1) * TG * NC * MT * 1024 + 256;
  if ((long long)NU9 * 256 * (q.SR > q.ext_h ? q.SR : q.ext_h) >= (1LL << 20)) return DPF_ERR_UNSUPPORTED;   // multiply-shift exactness
  const int sgn = d.transposed ? -1 : 1;
  const int t0 = d.transposed ? ((d.kd - 1) * d.dd * q.ext_h + (d.kh - 1) * dhl) * q.RS + (d.kw - 1) * d.dw : 0;
  for (int u = 0; u < 28; ++u) {
    const int c = u % d.kw, b2 = (u / d.kw) % d.kh, a = u / (d.kw * d.kh);
    q.tapoff[u] = u < T ? t0 + sgn * ((a * d.dd * q.ext_h + b2 * dhl) * q.RS + c * d.dw) : 0;
  }
  q.nchunks = (d.C + CC9 - 1) / CC9;
  q.tilesH = dpf_div_up(d.OH, q.thp * q.rstep) * q.rstep;
  q.tilesW = dpf_div_up(d.OW, 32);
  q.mSR = magic20(q.SR); q.mEH = magic20(q.ext_h);
  const long long nt9 = (long long)d.N * q.odt * q.tilesH * q.tilesW;
  if (nt9 <= 0 || nt9 > 0x3fffffffLL) return DPF_ERR_INVALID_ARG;
  q.ntiles = (int)nt9;
  q.cpx = (int)((nt9 + 7) / 8);
  q.stats = nullptr;
  if (stats) {
    if (nt9 * d.K * 2 > stats->capacity_doubles) return DPF_ERR_UNSUPPORTED;
    if (k_off > 0 && stats->parts != (int)nt9) return DPF_ERR_LAUNCH;   // a later slice must number the slab rows as the first one did
    q.stats = stats->slab; q.statsK = d.K; q.statsk0 = k_off;
  }
  unsigned short* wp = reinterpret_cast<unsigned short*>(ws);
  const long long total = (long long)q.nchunks * TG * MT * 512;
  // (f16 components: the rows' exponents (32 MT ints) sit behind the packed weights -- inside the three-component capacity of the workspace)
  int* wexp = reinterpret_cast<int*>(wp + ((total * NC + 7) & ~7LL));
  q.wexp = wexp;
  q.guard = dpf_h3_range_guard();
  if (NC == 2)
    hipLaunchKernelGGL(igemm3_pack_x9h_kernel, dim3(32 * MT), dim3(256), 0, st, w, wp, d.wA, d.wB, T, TG, MT, CC9, q.nchunks, d.mode, q.k0, kn, d.C, wexp);
  else
    hipLaunchKernelGGL(igemm3_pack_x9_kernel, dim3(dpf_ew_grid(total)), dim3(256), 0, st, w, wp, d.wA, d.wB, T, TG, MT, CC9, NC, q.nchunks, d.mode, q.k0, kn,
                       d.C);
  if (dpf_check_launch() != DPF_OK) return DPF_ERR_LAUNCH;
  if (stats) stats->parts = (int)nt9;
  const long long blocks9 = 8LL * q.cpx;
#define X9L(M, N_, C_)                                                                                                   \
  do {                                                                                                                 \
    if (NC == 3) return sh ? launch_x9<M, N_, C_, true, 3>(x, wp, bias, out, q, lds9, blocks9, st)                      \
                           : launch_x9<M, N_, C_, false, 3>(x, wp, bias, out, q, lds9, blocks9, st);                     \
    if (NC == 2) return sh ? launch_x9<M, N_, C_, true, 2>(x, wp, bias, out, q, lds9, blocks9, st)                      \
                           : launch_x9<M, N_, C_, false, 2>(x, wp, bias, out, q, lds9, blocks9, st);                     \
    return sh ? launch_x9<M, N_, C_, true, 1>(x, wp, bias, out, q, lds9, blocks9, st)                                   \
              : launch_x9<M, N_, C_, false, 1>(x, wp, bias, out, q, lds9, blocks9, st);                                  \
  } while (0)
  if (CC9 == 4) {
    if (MT == 1) { if (NT == 4) X9L(1, 4, 4); X9L(1, 2, 4); }
    X9L(2, 2, 4);
  }
  if (MT == 1) { if (NT == 4) X9L(1, 4, 8); X9L(1, 2, 8); }
  X9L(2, 2, 8);
#undef X9L
}
}  // namespace


namespace {
template <int CC, bool BF = false>
int launch_t2(const float* x, const float* wpk, const float* bias, float* out, const T2P& p, size_t lds, hipStream_t st) {
  static bool done = false;
  if (lds > 48 * 1024 && !done) {
    if (hipFuncSetAttribute((const void*)igemm2_tr2_kernel<CC, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DPF_ERR_LAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((igemm2_tr2_kernel<CC, BF>), dim3(8u * p.cpx), dim3(256), lds, st, x, wpk, bias, out, p);
  return dpf_check_launch();
}

// stride-2 transposed 3x3x3 (pad 1): loops over 32-channel output slices
int igemm2_tr2(const float* x, const float* w, const float* bias, float* out, float* ws, const DpfConvDesc& d, hipStream_t st) {
  static const int enabled = env_int("DPF_IGEMM2_TR2", 1), cc_over = 0;
  if (!enabled) return DPF_ERR_UNSUPPORTED;
  if (d.kd != 3 || d.kh != 3 || d.kw != 3 || d.sd != 2 || d.sh != 2 || d.sw != 2 || d.pd != 1 || d.ph != 1 || d.pw != 1 || d.dd != 1 ||
      d.dh != 1 || d.dw != 1)
    return DPF_ERR_UNSUPPORTED;
  // output extents an input of (ID, IH, IW) can reach: o = 2q + r <= 2*(I-1) + 1; larger outputs (not produced by these layers) -> generic kernel
  if (d.OD > 2 * d.ID || d.OH > 2 * d.IH || d.OW > 2 * d.IW) return DPF_ERR_UNSUPPORTED;
  const long long x_chan = (long long)d.ID * d.IH * d.IW;
  if (9 * x_chan >= (1LL << 30)) return DPF_ERR_UNSUPPORTED;
  const bool bf = dpf_conv_operand_bf16() != 0;
  const int CC = bf ? 8 : ((cc_over == 4 || cc_over == 8 || cc_over == 2) ? cc_over : 8);
  T2P p{};
  p.N = d.N; p.C = d.C; p.Ktot = d.Ktot;
  p.ID = d.ID; p.IH = d.IH; p.IW = d.IW; p.OD = d.OD; p.OH = d.OH; p.OW = d.OW;
  p.RS = 36; p.SR = 9; p.rpc = 2 * 5; p.chanStride = p.rpc * p.RS;
  p.nseg = CC * p.rpc * p.SR;
  p.nwseg = bf ? 27 * 32 : 27 * CC * 32 / 4;
  p.nchunks = (d.C + CC - 1) / CC;
  p.QD = dpf_div_up(d.OD, 2);
  p.tilesH = dpf_div_up(dpf_div_up(d.OH, 2), 4);
  p.tilesW = dpf_div_up(dpf_div_up(d.OW, 2), 32);
  const long long ntiles = (long long)d.N * p.QD * p.tilesH * p.tilesW;
  if (ntiles <= 0 || ntiles > 0x3fffffffLL) return DPF_ERR_INVALID_ARG;
  p.ntiles = (int)ntiles;
  p.cpx = (int)((ntiles + 7) / 8);
  p.mSR = magic20(p.SR); p.mRPC = magic20(p.rpc); p.mEH = magic20(5);
  p.vec = env_int("DPF_G2_VEC_STORE", 1);
  const size_t lds = 2 * (size_t)(CC * p.chanStride + 4 * p.nwseg) * sizeof(float);
  for (int k0 = 0; k0 < d.K; k0 += 32) {
    const int Kc = d.K - k0 < 32 ? d.K - k0 : 32;
    p.K = Kc; p.k0 = d.k0 + k0;
    const long long total = (long long)p.nchunks * 27 * CC * 32 + ZPAGE;
    if (bf)
      hipLaunchKernelGGL(igemm2_pack_bf16_kernel, dim3(dpf_ew_grid(total)), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(ws), d.wA, d.wB, 27,
                         32, p.nchunks, d.mode, p.k0, Kc, d.C);
    else
      hipLaunchKernelGGL(igemm2_pack_kernel, dim3(dpf_ew_grid(total)), dim3(256), 0, st, w, ws, d.wA, d.wB, 27, 32, CC, p.nchunks, d.mode, p.k0, Kc, d.C);
    if (dpf_check_launch() != DPF_OK) return DPF_ERR_LAUNCH;
    int rc;
    if (bf) {
      rc = launch_t2<8, true>(x, ws, bias, out, p, lds, st);
      if (rc != DPF_OK) return rc;
      continue;
    }
    switch (CC) { case 2: rc = launch_t2<2>(x, ws, bias, out, p, lds, st); break; case 4: rc = launch_t2<4>(x, ws, bias, out, p, lds, st); break;
                  default: rc = launch_t2<8>(x, ws, bias, out, p, lds, st); break; }
    if (rc != DPF_OK) return rc;
  }
  return DPF_OK;
}
}  // namespace

long long dpf_igemm2_workspace_floats(int T, int reduce, int outc) {
  const int KT = 32 * (((outc < 128 ? outc : 128) + 31) / 32);
  const long long plain = (long long)T * (reduce + 8) * KT + ZPAGE;
  // split bf16 fragments of the x9 kernel: 1 KB per (chunk, tap group, component, row tile) in either chunk layout
  const long long g4 = (long long)((reduce + 3) / 4) * ((T + 3) / 4), g8 = (long long)((reduce + 7) / 8) * ((T + 1) / 2);
  const long long x9 = (g4 > g8 ? g4 : g8) * 3 * (KT / 32) * 256;
  return plain > x9 ? plain : x9;
}

int dpf_igemm2_conv(const float* x, const float* w, const float* bias, float* out, float* ws, const DpfConvDesc& d, hipStream_t st,
                    DpfConvStats* stats) {
  static const int enabled = env_int("DPF_IGEMM2", 1);
  if (!enabled) return DPF_ERR_UNSUPPORTED;
  const int T = d.kd * d.kh * d.kw;
  if (T > MAXT || d.K > 128) return DPF_ERR_UNSUPPORTED;
  if ((d.IW & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(ws) & 15)) return DPF_ERR_UNSUPPORTED;
  const long long x_chan = (long long)d.ID * d.IH * d.IW;
  if (9 * x_chan >= (1LL << 30)) return DPF_ERR_UNSUPPORTED;      // per-lane 32-bit source offsets within a chunk
  if (T == 1 && !env_int("DPF_IGEMM2_1x1", 1)) return DPF_ERR_UNSUPPORTED;   // pointwise convs (HBM-bound): 1.8x the generic kernel
  if (stats && (d.transposed || d.Ktot != d.K)) return DPF_ERR_UNSUPPORTED;
  if (d.accumulate && d.transposed && (d.sd != 1 || d.sh != 1 || d.sw != 1)) return DPF_ERR_UNSUPPORTED;   // class-fused kernel: plain stores
  if (d.transposed && (d.sd != 1 || d.sh != 1 || d.sw != 1)) return igemm2_tr2(x, w, bias, out, ws, d, st);

  const int MT = (d.K + 31) / 32, KT = 32 * MT;
  static const int nt_over = 0, cc_over = 0, lds_target = 53 * 1024;
  int NT = MT == 1 ? 4 : 2;
  if (nt_over == 2 || (nt_over == 4 && MT <= 2)) NT = nt_over;
  // operand precision "bf16" (dpf_set_conv_operand_precision): 8-channel chunks; 16 position rows per workgroup when two buffers
  // of that patch fit the LDS, else 8
  bool bf = dpf_conv_operand_bf16() != 0 && T > 1;
  const bool bf_mode = bf;                                        // (bf is cleared below when the igemm2 bf16 kernel cannot take the shape)
  static const int bf3_on = env_int("DPF_IGEMM3_BF", 1);
  int p_single = 0;
  if (bf) {
    auto patch_for = [&](int nt) {      // bytes of an 8-channel fp32 patch
      const int sxh = d.transposed ? 1 : d.sh, sxw = d.transposed ? 1 : d.sw;
      const int e0w = d.transposed ? d.pw - (d.kw - 1) * d.dw : -d.pw;
      const int ext_d = (d.kd - 1) * d.dd + 1, ext_h = (4 * nt - 1) * sxh + (d.kh - 1) * d.dh + 1, ext_w = 31 * sxw + (d.kw - 1) * d.dw + 1;
      const int rs = ((((e0w % 4) + 4) % 4 + ext_w + 3) / 4) * 4;
      return (long long)8 * ext_d * ext_h * rs * 4;
    };
    const long long wbytes = (long long)T * KT * 16;
    // (rows per workgroup, one or two LDS buffers): the candidate with the most resident workgroups per CU wins (LDS and the
    // register budget of the instantiation; more than 3 buys nothing) -- at equal residency two buffers beat one and 16 rows beat 8
    // (less halo).  tools/conv_bf16_bench.py: the 3-D K = 32 convs run 2x faster on 8 rows x 1 buffer (3 resident) than on 2 buffers.
    static const int single_over = -1;
    auto occ_regs = [&](int nt) {
      const int est = MT * nt * 16 + 2 * (2 * MT + 4 * nt) + 2 * NLD + 44;
      return est <= 128 ? 4 : (est <= 168 ? 3 : 2);
    };
    int best = -1;
    for (int nt : {4, 2}) {
      if ((nt == 4 && MT > 2) || patch_for(nt) > 2LL * NLD * 256 * 16) continue;
      if ((nt_over == 2 || nt_over == 4) && nt != nt_over) continue;
      for (int single = 0; single < 2; ++single) {
        if (single_over >= 0 && single != single_over) continue;
        const long long lds_c = (single ? 1 : 2) * (patch_for(nt) + wbytes);
        if (lds_c > 160 * 1024) continue;
        int resid = (int)(160 * 1024 / lds_c);
        if (resid > occ_regs(nt)) resid = occ_regs(nt);
        if (resid > 3) resid = 3;
        const int score = resid * 4 + (single ? 0 : 2) + (nt == 4 ? 1 : 0);
        if (score > best) { best = score; NT = nt; p_single = single; }
      }
    }
    if (best < 0) { bf = false; p_single = 0; }     // e.g. stride-2 forward patches: exact-f32 kernel below
    if (!bf) { NT = MT == 1 ? 4 : 2; if (nt_over == 2 || (nt_over == 4 && MT <= 2)) NT = nt_over; }
  }
  const int TH = 4 * NT;

  G2P p{};
  p.N = d.N; p.C = d.C; p.K = d.K; p.Ktot = d.Ktot; p.k0 = d.k0;
  p.ID = d.ID; p.IH = d.IH; p.IW = d.IW; p.OD = d.OD; p.OH = d.OH; p.OW = d.OW;
  p.T = T;
  p.accum = d.accumulate;
  p.rstep = 1;
  static const int vec_on = env_int("DPF_G2_VEC_STORE", 1);
  p.vec = vec_on && (d.OW & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  int ext_w;
  if (!d.transposed) {
    p.sxd = d.sd; p.sxh = d.sh; p.sxw = d.sw;
    p.e0d = -d.pd; p.e0h = -d.ph; p.e0w = -d.pw;
  } else {
    p.sxd = p.sxh = p.sxw = 1;
    p.e0d = d.pd - (d.kd - 1) * d.dd; p.e0h = d.ph - (d.kh - 1) * d.dh; p.e0w = d.pw - (d.kw - 1) * d.dw;
  }
  ext_w = 31 * p.sxw + (d.kw - 1) * d.dw + 1;
  p.colshift = ((p.e0w % 4) + 4) % 4;
  p.RS = ((p.colshift + ext_w + 3) / 4) * 4;
  // depth tiles: pz output planes x TH / pz rows.  A 3-plane kernel stages kd - 1 halo planes per tile, so one output plane per tile
  // fetches every input plane three times (16 x 32 tile: 3 x 18 = 54 patch rows per 16 position rows; 4 planes x 4 rows: 6 x 6 = 36).
  // The split with the fewest staged rows per position row wins; DPF_G2_PZ = 1 | 2 | 4 forces one.
  {
    static const int pz_over = env_int("DPF_G2_PZ", 0);
    auto pz_ok = [&](int pz) { return pz == 1 || (d.kd > 1 && pz <= TH / 2 && pz <= d.OD); };
    auto pz_cost = [&](int pz) {
      const int thp = TH / pz;
      return (double)((pz - 1) * p.sxd + (d.kd - 1) * d.dd + 1) * ((thp - 1) * p.sxh + (d.kh - 1) * d.dh + 1) / (double)(pz * thp);
    };
    int best_pz = 1;
    for (int pz : {2, 4})
      if (pz_ok(pz) && pz_cost(pz) < pz_cost(best_pz) - 1e-9) best_pz = pz;
    if ((pz_over == 1 || pz_over == 2 || pz_over == 4) && pz_ok(pz_over)) best_pz = pz_over;
    p.pz = best_pz;
    p.thp = TH / p.pz;
    p.thp_shift = 0;
    while ((1 << p.thp_shift) < p.thp) ++p.thp_shift;
    p.odt = dpf_div_up(d.OD, p.pz);
  }
  p.ext_d = (p.pz - 1) * p.sxd + (d.kd - 1) * d.dd + 1;
  p.ext_h = (p.thp - 1) * p.sxh + (d.kh - 1) * d.dh + 1;
  p.planeStride = p.ext_h * p.RS;
  p.SR = p.RS / 4;
  p.rpc = p.ext_d * p.ext_h;
  p.chanStride = p.rpc * p.RS;
  {
    const int sgn = d.transposed ? -1 : 1;
    const int stepA = sgn * d.dd * p.ext_h * p.RS, stepB = sgn * d.dh * p.RS;
    p.stepC = sgn * d.dw;
    p.incB = stepB - (d.kw - 1) * p.stepC;
    p.incA = stepA - (d.kh - 1) * stepB - (d.kw - 1) * p.stepC;
    p.steps = 0;
    for (int t = 0; t < T; ++t) {
      const int c = t % d.kw, b = (t / d.kw) % d.kh;
      const unsigned long long code = t == T - 1 ? 3 : (c + 1 < d.kw ? 0 : (b + 1 < d.kh ? 1 : 2));
      p.steps |= code << (2 * t);
    }
    p.tap0 = d.transposed ? ((d.kd - 1) * d.dd * p.ext_h + (d.kh - 1) * d.dh) * p.RS + (d.kw - 1) * d.dw : 0;
  }
  // ---- igemm3_x9_kernel: stride-1 launches, output channels in slices of at most 64 -- fp32 products from exact bf16 splits, or (operand
  //      precision "bf16") the operands rounded to bf16
  // (bf16 operands, 2-D kernels with more than 64 output channels: igemm2's bf16 kernel stages the patch once for all of them and is faster)
  if (T > 4 && p.sxd == 1 && p.sxh == 1 && p.sxw == 1 && (!bf_mode || (bf3_on && !(d.K > 64 && d.kd == 1)))) {
    const int nc = bf_mode ? 1 : dpf_conv_f32_nc();
    if (d.K <= 64) {
      constexpr int nt3 = 0;       // (2 = 8-row tiles for <= 32 output channels too, three resident workgroups: measured slower)
      const int rc = x9_try(x, w, bias, out, ws, d, p, 0, d.K, MT == 1 ? ((nt3 == 2 || (nt3 == 12 && d.kd == 1)) ? 2 : 4) : 2, stats, st, nc);
      if (rc != DPF_ERR_UNSUPPORTED) return rc;
    } else {
      for (int k_off = 0; k_off < d.K; k_off += 64) {
        const int rc = x9_try(x, w, bias, out, ws, d, p, k_off, d.K - k_off < 64 ? d.K - k_off : 64, 2, stats, st, nc, 2);
        if (rc == DPF_ERR_UNSUPPORTED && k_off == 0) break;      // nothing launched yet: the kernels below take the launch
        if (rc != DPF_OK) return rc == DPF_ERR_UNSUPPORTED ? DPF_ERR_LAUNCH : rc;
        if (k_off + 64 >= d.K) return DPF_OK;
      }
    }
  }
  // channels per chunk: the largest of {8, 4, 2} whose two buffers leave room for >= 3 resident workgroups
  auto buf_bytes = [&](int cc) { return (size_t)(cc * p.chanStride + T * cc * KT) * sizeof(float); };
  auto nseg_of = [&](int cc) { return cc * p.rpc * p.SR; };
  // Launches that fill the chip several times over run best with the smallest chunk (more resident workgroups hide the DMA
  // latency: +1..5 % on every large shape, tools/conv_shape_bench.py sweep); small launches (< 2 tiles per CU) keep the larger
  // chunks, which shorten their few workgroups' barrier chains.
  const long long ntiles_est = (long long)d.N * p.odt * dpf_div_up(d.OH, p.thp) * dpf_div_up(d.OW, 32);
  int CC = 0;
  if (ntiles_est >= 512 && 2 * buf_bytes(2) <= (size_t)lds_target && nseg_of(2) <= NLD * 256) CC = 2;
  if (!CC)
    for (int cc : {8, 4, 2})
      if (2 * buf_bytes(cc) <= (size_t)lds_target && nseg_of(cc) <= NLD * 256) { CC = cc; break; }
  if (cc_over == 2 || cc_over == 4 || cc_over == 8) CC = cc_over;
  if (!CC) CC = 2;
  if (bf) CC = 8;
  const size_t bf_buf = (size_t)8 * p.chanStride * sizeof(float) + (size_t)T * KT * 16;
  if (bf ? (nseg_of(8) > 2 * NLD * 256 || 2 * bf_buf > 160 * 1024) : (nseg_of(CC) > NLD * 256 || 2 * buf_bytes(CC) > 160 * 1024))
    return DPF_ERR_UNSUPPORTED;
  p.nseg = nseg_of(CC);
  p.nwseg = bf ? T * KT : T * CC * KT / 4;
  p.nchunks = (d.C + CC - 1) / CC;
  p.tilesH = dpf_div_up(d.OH, p.thp);
  p.tilesW = dpf_div_up(d.OW, 32);
  p.mSR = magic20(p.SR); p.mRPC = magic20(p.rpc); p.mEH = magic20(p.ext_h);
  if ((long long)(bf ? 2 : 1) * NLD * 256 * (p.SR > p.rpc ? p.SR : p.rpc) >= (1LL << 20)) return DPF_ERR_UNSUPPORTED;   // multiply-shift exactness
  const long long ntiles = (long long)d.N * p.odt * p.tilesH * p.tilesW;
  if (ntiles <= 0 || ntiles > 0x3fffffffLL) return DPF_ERR_INVALID_ARG;
  p.ntiles = (int)ntiles;
  p.cpx = (int)((ntiles + 7) / 8);
  const long long blocks = 8LL * p.cpx;

  const long long total = (long long)p.nchunks * T * CC * KT + ZPAGE;
  if (bf)
    hipLaunchKernelGGL(igemm2_pack_bf16_kernel, dim3(dpf_ew_grid(total + ZPAGE)), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(ws), d.wA,
                       d.wB, T, KT, p.nchunks, d.mode, d.k0, d.K, d.C);
  else
    hipLaunchKernelGGL(igemm2_pack_kernel, dim3(dpf_ew_grid(total)), dim3(256), 0, st, w, ws, d.wA, d.wB, T, KT, CC, p.nchunks, d.mode, d.k0, d.K, d.C);
  if (dpf_check_launch() != DPF_OK) return DPF_ERR_LAUNCH;

  p.single = p_single;
  const size_t lds = bf ? (p_single ? 1 : 2) * bf_buf : 2 * buf_bytes(CC);
  p.stats = nullptr;
  if (stats) {
    if (ntiles * d.K * 2 > stats->capacity_doubles || lds < (size_t)4 * 2 * MT * 16 * 2 * sizeof(double)) return DPF_ERR_UNSUPPORTED;
    p.stats = stats->slab;
    p.statsK = d.K; p.statsk0 = 0;
    stats->parts = (int)ntiles;
  }
#define G2B(M, N_) return launch_g2<M, N_, 8, true>(x, ws, bias, out, p, lds, blocks, st)
  if (bf) {
    if (NT == 4) {
      if (MT == 1) { G2B(1, 4); } else { G2B(2, 4); }
    }
    switch (MT) {
      case 1: G2B(1, 2);
      case 2: G2B(2, 2);
      case 3: G2B(3, 2);
      default: G2B(4, 2);
    }
  }
#undef G2B
#define G2(M, N_, C_) return launch_g2<M, N_, C_>(x, ws, bias, out, p, lds, blocks, st)
#define G2CC(M, N_)                                                                                                         \
  switch (CC) { case 8: G2(M, N_, 8); case 4: G2(M, N_, 4); default: G2(M, N_, 2); }
  if (NT == 4) {
    if (MT == 1) { G2CC(1, 4) } else { G2CC(2, 4) }
  }
  switch (MT) {
    case 1: G2CC(1, 2)
    case 2: G2CC(2, 2)
    case 3: G2CC(3, 2)
    default: G2CC(4, 2)
  }
#undef G2CC
#undef G2
}

#ifdef DPF_STAMPS
extern "C" int dpf_debug_x9_passes(unsigned long long* host_out4) {      // reads and clears the pass counters
  const unsigned long long z[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(host_out4, HIP_SYMBOL(g_x9_passes), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_x9_passes), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" int dpf_debug_x9_stamps(unsigned long long* host_out, int nblocks) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_x9_stamps), sizeof(unsigned long long) * 8 * (size_t)nblocks) == hipSuccess ? 0 : -1;
}
#endif
