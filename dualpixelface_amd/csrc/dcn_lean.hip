// Deformable 3-D convolution, "lean sampler" kernels for the configuration StereoDPNet runs (normal_module.py:14-19: 3x3x3 taps,
// stride 1, padding 1, dilation 1, depth 4): same arithmetic as dcn3d.hip (reference: src/module/dcn3d/src/cuda/deform_im2col_cuda.cuh:26-72
// sampler, :192-265 im2col + validity rule :248; deform_conv_cuda.cu:93-123 GEMM + bias), different instruction budget.
//
// What the measurements said (tools/lean_probe.hip, tools/lean_probe2.hip):
//   * round 3's role-split kernels were bound by the SAMPLER's instruction stream -- ~700 vector instructions per (64 voxels x 16 channels
//     x 1 tap), 128 of them the essential FMAs -- while the matrix waves need 2048 clocks for the same step;
//   * on this chip an fp32 MFMA and vector-ALU work never overlap on a SIMD -- not across waves (a wave issuing MFMAs back to back starves
//     its SIMD partner of vector and LDS issue) and not inside a wave (4 v_pk_fma_f32 behind a 64-clock MFMA: +34 clocks) -- but LDS reads a
//     wave issues between its OWN MFMAs are nearly free (2 ds_read_b128 per MFMA: +7 clocks).
// The diet:
//   * quad-planar LDS image  region[q][cell][4 channels]  (q = channel quad): the 16 lanes of a ds_read_b128 group read 16 consecutive
//     cells = 256 contiguous bytes (conflict free without a swizzle), and because RY / RX are template constants every corner of every
//     quad is  base + immediate offset : two address registers per tap instead of 32 computed addresses;
//   * the region is staged WITH its out-of-volume cells as zeros, so corners outside the volume along y / x read zeros and need no masks
//     (cuh:43-65 returns 0 for them); only z (whole depth staged, D <= 4) keeps its two masks; the validity rule cuh:248 is kept explicitly
//     (a sample exactly on -1 must not hand out a coordinate derivative);
//   * the lane -> voxel map is permuted so that each lane group of a ds_read_b128 holds 16 x-consecutive voxels;
//   * trilinear weights and the 8 x CH multiply-adds on v_pk_mul_f32 / v_pk_fma_f32 with op_sel broadcasting the weight (no splat moves);
//   * forward: every wave samples AND contracts its own 64 voxels -- corner reads of tap t + 1 in the shadow of the MFMAs of tap t, B operands
//     from the wave's own registers (v_permlane32_swap_b32), no sample tile, no barrier inside a chunk, two workgroups per CU;
//   * grad_offset + grad_weight: 4 sampler waves (a voxel per lane, all CH channels) + 4 matrix waves that run both products;
//   * a sample whose corner block leaves the staged box is redone by the whole wave (lane = channel x corner pair: two global loads per
//     lane, one round trip) instead of 16 serial channel round trips in the one slow lane.
#include "dcn_internal.h"
#include "conv_internal.h"
#include <cstdlib>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct LeanP {
  int B, C, K, D, H, W;
  int Cpad, KT;
  long long P;          // D * H * W (= output voxels per channel: stride 1, padding 1)
  int tilesY, tilesX;   // a tile spans the whole depth
  int nchunk;           // channel chunks
};

// XCD-aware tile order (see dcn3d.hip: dpf_xcd_tile)
__device__ __forceinline__ int lean_xcd_tile(int blk, int n) {
  const int q = n >> 3, r = n & 7, x = blk & 7, i = blk >> 3;
  return x * q + (x < r ? x : r) + i;
}

template <int CH_, int TY_, int TX_, int RYH_, int RXL_, int RXR_>
struct Geo {
  static constexpr int CH = CH_, NQ = CH_ / 4, TY = TY_, TX = TX_, RYH = RYH_, RXL = RXL_;
  static constexpr int NV = 4 * TY * TX;                       // output voxels per workgroup (4 depth planes)
  static constexpr int RY = TY + 2 + 2 * RYH;
  static constexpr int RX = (TX + 2 + RXL + RXR_ + 3) / 4 * 4;
  static constexpr int ZS = RY * RX * 16;                      // bytes between depth planes of a quad plane
  static constexpr int PLANE = 4 * ZS;                         // bytes of one quad plane of the region image
  static constexpr int SQ = NV * 16;                           // bytes of one quad plane of the sample tile
  static constexpr int SBUF = NQ * SQ;                         // one sample tile
  static constexpr int LDS = NQ * PLANE + 2 * SBUF;
  static constexpr int NS = NV / 64;                           // sampler waves (= matrix waves)
  static_assert((1 + RXL) % 4 == 0 && TX % 16 == 0 && NV % 64 == 0, "aligned region origin / whole waves");
};

// lane (0..31) -> position (0..31) such that the two 16-lane groups ds_read_b128 services together ({0-3,12-15,20-27}, {4-11,16-19,28-31}:
// MI355X_MICROARCH.md, LDS table) hold positions 0..15 and 16..31
__device__ __forceinline__ int lane_pos32(int l) { return (int)((0x73261540u >> (4 * (l >> 2))) & 7u) * 4 + (l & 3); }

__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 w, f32x2 v) {   // w.x * v
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(v));
  return r;
}
__device__ __forceinline__ f32x2 pk_mul_hi(f32x2 w, f32x2 v) {   // w.y * v
  f32x2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "v"(v));
  return r;
}
__device__ __forceinline__ void pk_fma_lo(f32x2& acc, f32x2 w, f32x2 v) {   // acc += w.x * v
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(v));
}
__device__ __forceinline__ void pk_fma_hi(f32x2& acc, f32x2 w, f32x2 v) {   // acc += w.y * v
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(v));
}

// Stage x[b, c0 .. c0 + CH) over the region rows [ry0, ry0 + RY) x columns [rx0, rx0 + RX) x all D planes into region[q][cell][4];
// cells outside the volume and channels beyond C are zeros.  A unit = 4 channels x 4 consecutive x: four float4 loads, a register
// transpose, four ds_write_b128 into 64 consecutive bytes.
template <class G>
__device__ __forceinline__ void lean_stage(const LeanP& p, const float* __restrict__ xb, int c0, char* region, int ry0, int rx0, int tid,
                                           int nthreads) {
  constexpr int SR = G::RX / 4;
  const long long chan = p.P;
  const int units = p.D * G::RY * G::NQ * SR;
  for (int u = tid; u < units; u += nthreads) {
    const int seg = u % SR;
    const int it = u / SR;
    const int cg = it % G::NQ;
    const int row = it / G::NQ;
    const int lz = row / G::RY, ly = row - lz * G::RY;
    const int gy = ry0 + ly, gx = rx0 + 4 * seg;
    const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const float* src = xb + (long long)(c0 + 4 * cg) * chan + ((long long)lz * p.H + gy) * p.W + gx;
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (ok && c0 + 4 * cg + q < p.C) ? *reinterpret_cast<const f32x4*>(src + q * chan) : f32x4{0.f, 0.f, 0.f, 0.f};
    char* dst = region + cg * G::PLANE + (row * G::RX + 4 * seg) * 16;
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0].x, v[1].x, v[2].x, v[3].x};
    *reinterpret_cast<f32x4*>(dst + 16) = f32x4{v[0].y, v[1].y, v[2].y, v[3].y};
    *reinterpret_cast<f32x4*>(dst + 32) = f32x4{v[0].z, v[1].z, v[2].z, v[3].z};
    *reinterpret_cast<f32x4*>(dst + 48) = f32x4{v[0].w, v[1].w, v[2].w, v[3].w};
  }
}

#ifdef DPF_STAMPS
__device__ unsigned long long g_lean_stamps[16 * 128 * 4];
#define LEAN_STAMP4(step, slot) \
  if (blockIdx.x == 3000 && lane == 0 && (step) < 128) g_lean_stamps[((wave_u + 8) * 128 + (step)) * 4 + (slot)] = __builtin_readcyclecounter();
#else
#define LEAN_STAMP4(step, slot)
#endif

// Per (voxel, tap) table of the sampler: where the 2 x 2 x 2 corner block sits in the staged region and the eight trilinear weights.
struct LeanTab {
  int a0, a1;                   // byte offsets of the (low y, low x) corner cell on the low / high depth plane
  f32x2 w00, w01, w10, w11;     // weights of (jd, jh) = (0,0), (0,1), (1,0), (1,1), each as the pair (jw = 0, jw = 1)
  int d0, h0, w0;               // kept for the rare sample that leaves the staged box
  float ld, lh, lw;
  bool inreg;                   // corner block inside the staged box
  bool slow;                    // a sample the fast path cannot serve: voxel inside the tile's volume part, sample valid (cuh:248), box left
  f32x2 wz, wy, wx;             // axis factors (low, high); wz carries the in-volume mask of its plane and the validity of the sample
  f32x2 mz;                     // 1 / 0: depth plane inside the volume and sample usable (coordinate derivative along z)
};

template <class G>
__device__ __forceinline__ LeanTab lean_tab(const LeanP& p, bool pvalid, int ry0, int rx0, float fd, float fh, float fw) {
  LeanTab t;
  const float d0f = floorf(fd), h0f = floorf(fh), w0f = floorf(fw);
  t.ld = fd - d0f; t.lh = fh - h0f; t.lw = fw - w0f;
  t.d0 = (int)d0f; t.h0 = (int)h0f; t.w0 = (int)w0f;
  const int ly = t.h0 - ry0, lx = t.w0 - rx0;
  t.inreg = (unsigned)ly < (unsigned)(G::RY - 1) && (unsigned)lx < (unsigned)(G::RX - 1);
  // cuh:248: a sample outside (-1, D) x (-1, H) x (-1, W) contributes nothing, not even through its coordinate derivative (the zero-padded
  // image alone would still hand a derivative to a sample sitting exactly on -1)
  const bool valid = pvalid && fd > -1.f && fh > -1.f && fw > -1.f && fd < (float)p.D && fh < (float)p.H && fw < (float)p.W;
  const bool ok = t.inreg && valid;
  t.slow = valid && !t.inreg;
  // z: the whole depth is staged; planes outside the volume get weight 0 and a clamped index
  const bool mz0 = (unsigned)t.d0 < (unsigned)p.D, mz1 = (unsigned)(t.d0 + 1) < (unsigned)p.D;
  f32x2 wz, wy, wx;
  wz.x = (ok && mz0) ? 1.f - t.ld : 0.f;
  wz.y = (ok && mz1) ? t.ld : 0.f;
  wy.x = 1.f - t.lh; wy.y = t.lh;
  wx.x = 1.f - t.lw; wx.y = t.lw;
  t.wz = wz; t.wy = wy; t.wx = wx;
  t.mz.x = (ok && mz0) ? 1.f : 0.f; t.mz.y = (ok && mz1) ? 1.f : 0.f;
  const int Dm1 = p.D - 1;
  const int iz0 = min(max(t.d0, 0), Dm1), iz1 = min(max(t.d0 + 1, 0), Dm1);
  const int cell = ok ? __mul24(__mul24(iz0, G::RY) + ly, G::RX) + lx : 0;
  t.a0 = cell * 16;
  t.a1 = t.a0 + __mul24(iz1 - iz0, G::ZS);
  const f32x2 zy0 = pk_mul_lo(wz, wy), zy1 = pk_mul_hi(wz, wy);          // (wz0 wy0, wz0 wy1), (wz1 wy0, wz1 wy1)
  t.w00 = pk_mul_lo(zy0, wx); t.w01 = pk_mul_hi(zy0, wx);
  t.w10 = pk_mul_lo(zy1, wx); t.w11 = pk_mul_hi(zy1, wx);
  return t;
}

#define LEAN_LOAD8(dst, q)                                                                        \
  dst[0] = *reinterpret_cast<const f32x4*>(r0 + (q) * G::PLANE);                                  \
  dst[1] = *reinterpret_cast<const f32x4*>(r0 + (q) * G::PLANE + 16);                             \
  dst[2] = *reinterpret_cast<const f32x4*>(r0 + (q) * G::PLANE + G::RX * 16);                     \
  dst[3] = *reinterpret_cast<const f32x4*>(r0 + (q) * G::PLANE + G::RX * 16 + 16);                \
  dst[4] = *reinterpret_cast<const f32x4*>(r1 + (q) * G::PLANE);                                  \
  dst[5] = *reinterpret_cast<const f32x4*>(r1 + (q) * G::PLANE + 16);                             \
  dst[6] = *reinterpret_cast<const f32x4*>(r1 + (q) * G::PLANE + G::RX * 16);                     \
  dst[7] = *reinterpret_cast<const f32x4*>(r1 + (q) * G::PLANE + G::RX * 16 + 16);

// ---- forward, second form: every wave samples AND contracts its own 64 voxels.  Measured on this chip (tools/lean_probe.hip,
// tools/lean_probe2.hip): an fp32 MFMA and vector-ALU work never overlap on a SIMD -- not across waves (a wave issuing MFMAs back to back
// starves its SIMD partner of vector and LDS issue) and not inside one wave (4 v_pk_fma_f32 behind a 64-clock MFMA cost 34 clocks) -- but
// LDS reads issued by the SAME wave between its MFMAs are nearly free (2 ds_read_b128 per MFMA: +7 clocks).  So the role split bought no
// overlap (step = sampler alone + matrix wave alone, stamps in DESIGN section 4) while paying a barrier per tap and an LDS round trip of the
// samples.  Here a wave issues the corner reads of tap t + 1 in the shadow of its MFMAs of tap t, accumulates them afterwards, and turns
// its own samples into the B operands of both 32-voxel column tiles with v_permlane32_swap_b32 (channel pair (2s, 2s + 1) of the low half's
// voxels becomes k-step s of tile 0, of the high half's voxels of tile 1): no sample tile in LDS, no barrier inside a chunk.
template <int CH>
__global__ void lean_repack_fwd1_kernel(const float* __restrict__ w, float* __restrict__ wl, int K, int C, int MT, int nchunk) {
  // wl[tap][chunk][m][lane][8]: fragment s of lane (l31, hh) = W[k = 32 m + l31][c = chunk * CH + 2 s + hh][tap] (zero beyond K / C / CH)
  const int total = 27 * nchunk * MT * 512;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int s = i & 7, lane = (i >> 3) & 63;
    int r = i >> 9;
    const int m = r % MT; r /= MT;
    const int chunk = r % nchunk;
    const int t = r / nchunk;
    const int k = 32 * m + (lane & 31), cl = 2 * s + (lane >> 5), c = chunk * CH + cl;
    wl[i] = (cl < CH && k < K && c < C) ? w[((long long)k * C + c) * 27 + t] : 0.f;
  }
}

template <class G, int MT>
__global__ __launch_bounds__(256, 2) void dcn_lean_fwd1_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                            const float* __restrict__ wl, const float* __restrict__ bias,
                                                            float* __restrict__ out, LeanP p) {
  extern __shared__ __align__(16) char smem[];
  constexpr int CH = G::CH, NQ = G::NQ, KSTEPS = CH / 2, T = 27;
  static_assert(G::NV == 256, "four waves of 64 voxels");
  char* region = smem;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  int blk = lean_xcd_tile(blockIdx.x, gridDim.x);
  const int tx = blk % p.tilesX; blk /= p.tilesX;
  const int ty = blk % p.tilesY;
  const int b = blk / p.tilesY;
  const int y0 = ty * G::TY, x0 = tx * G::TX;
  const int ry0 = y0 - 1 - G::RYH, rx0 = x0 - 1 - G::RXL;
  const float* xb = x + (long long)b * p.C * p.P;
  const int vox = wave_u * 64 + (lane & 32) + lane_pos32(l31);
  const int px = vox % G::TX, py = (vox / G::TX) % G::TY, pz = vox / (G::TX * G::TY);
  const int zo = pz, yo = y0 + py, xo = x0 + px;
  const bool pvalid = zo < p.D && yo < p.H && xo < p.W;
  const long long ppos = pvalid ? ((long long)zo * p.H + yo) * p.W + xo : 0;
  const float* offp0 = offset + (long long)b * 3 * T * p.P + ppos;
  const float zbf = (float)(zo - 1), ybf = (float)(yo - 1), xbf = (float)(xo - 1);
  const long long P3 = 3 * p.P;
  const unsigned wlane = (unsigned)lane * 32u;
  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][nt][j] = 0.f;

  // accumulate two quads of a tap from their 8 corner reads each into 4 + 4 sample registers (four independent multiply-add chains)
  auto accum2 = [&](const LeanTab& tb, const f32x4* c, const f32x4* e, float* sv, float* su, bool both) {
    const f32x2 zy0 = pk_mul_lo(tb.wz, tb.wy), zy1 = pk_mul_hi(tb.wz, tb.wy);
    const f32x2 w00 = pk_mul_lo(zy0, tb.wx), w01 = pk_mul_hi(zy0, tb.wx), w10 = pk_mul_lo(zy1, tb.wx), w11 = pk_mul_hi(zy1, tb.wx);
    f32x2 lo = pk_mul_lo(w00, c[0].xy), hi = pk_mul_lo(w00, c[0].zw), l2 = {0.f, 0.f}, h2 = {0.f, 0.f};
    if (both) { l2 = pk_mul_lo(w00, e[0].xy); h2 = pk_mul_lo(w00, e[0].zw); }
#define LEAN_ACC(J, W, HL)                                                                    \
    pk_fma_##HL(lo, W, c[J].xy); pk_fma_##HL(hi, W, c[J].zw);                                 \
    if (both) { pk_fma_##HL(l2, W, e[J].xy); pk_fma_##HL(h2, W, e[J].zw); }
    LEAN_ACC(1, w00, hi) LEAN_ACC(2, w01, lo) LEAN_ACC(3, w01, hi) LEAN_ACC(4, w10, lo) LEAN_ACC(5, w10, hi) LEAN_ACC(6, w11, lo) LEAN_ACC(7, w11, hi)
#undef LEAN_ACC
    sv[0] = lo.x; sv[1] = lo.y; sv[2] = hi.x; sv[3] = hi.y;
    if (both) { su[0] = l2.x; su[1] = l2.y; su[2] = h2.x; su[3] = h2.y; }
  };
  // samples that leave the staged box: redone by the wave from global memory (lane = channel x corner pair), then handed to the owning lane
  auto slow_fix = [&](const LeanTab& tb, float* sv, int c0) {
    unsigned long long slow = __ballot(tb.slow);
    while (slow) {
      const int L = __builtin_ctzll(slow);
      slow &= slow - 1;
      const int sd0 = __builtin_amdgcn_readlane(tb.d0, L), sh0 = __builtin_amdgcn_readlane(tb.h0, L), sw0 = __builtin_amdgcn_readlane(tb.w0, L);
      const float sld = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb.ld), L));
      const float slh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb.lh), L));
      const float slw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb.lw), L));
      const int ch = lane & 15, jd = lane >> 5, jh = (lane >> 4) & 1;
      const int dz = sd0 + jd, hy = sh0 + jh;
      const int cgl = c0 + ch;
      const bool rowin = ch < CH && cgl < p.C && (unsigned)dz < (unsigned)p.D && (unsigned)hy < (unsigned)p.H;
      const bool in0 = rowin && (unsigned)sw0 < (unsigned)p.W, in1 = rowin && (unsigned)(sw0 + 1) < (unsigned)p.W;
      const float* xr = xb + (long long)(cgl < p.C ? cgl : 0) * p.P + ((long long)(rowin ? dz : 0) * p.H + (rowin ? hy : 0)) * p.W;
      const float v0 = in0 ? xr[sw0] : 0.f, v1 = in1 ? xr[sw0 + 1] : 0.f;
      const float wzy = (jd ? sld : 1.f - sld) * (jh ? slh : 1.f - slh);
      float part = fmaf(wzy * slw, v1, (wzy * (1.f - slw)) * v0);
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);                // lanes 0 .. 15 (and their copies) hold channel `lane & 15` of voxel L
#pragma unroll
      for (int ch2 = 0; ch2 < CH; ++ch2) {
        const float vch = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), ch2));
        if (lane == L) sv[ch2] = vch;
      }
    }
  };

  // offsets ring: slot (u % 3) holds tap u's three components, fetched four taps ahead
  float od[3], oh[3], ow[3];
  LeanTab tab = lean_tab<G>(p, pvalid, ry0, rx0, zbf + offp0[0], ybf + offp0[p.P], xbf + offp0[2 * p.P]);
#pragma unroll
  for (int u = 1; u <= 3; ++u) { od[u % 3] = offp0[u * P3]; oh[u % 3] = offp0[u * P3 + p.P]; ow[u % 3] = offp0[u * P3 + 2 * p.P]; }

  LeanTab tab1;                                        // tap 1's table while tap 0 is gathered at a chunk's start
  // one tap: contract tap t (samples in `cur`, weight fragments in `a`) while the corners of tap t + 1 are read; then accumulate tap t + 1
  // into `nxt`.  `tab` is tap t + 1's table on entry and tap t + 2's on exit.  last: tap t + 1 belongs to the next chunk (nothing to read).
  auto body = [&](int t, int chunk, int c0, float (&cur)[16], float (&nxt)[16], f32x4 (&a)[MT][2], f32x4 (&an)[MT][2], bool last) {
    // next tap's weight fragments (L2) and the tap after next's offsets (HBM)
    {
      const int tn = t + 1 < T ? t + 1 : 0, cn = t + 1 < T ? chunk : (chunk + 1 < p.nchunk ? chunk + 1 : chunk);
      const char* wb = reinterpret_cast<const char*>(wl) + ((long long)(tn * p.nchunk + cn) * MT) * 2048;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        an[m][0] = *reinterpret_cast<const f32x4*>(wb + wlane + m * 2048);
        an[m][1] = *reinterpret_cast<const f32x4*>(wb + wlane + m * 2048 + 16);
      }
    }
    // B operands of both column tiles from this wave's own samples
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(cur[2 * s]), "+v"(cur[2 * s + 1]));
    const char* r0 = region + tab.a0;
    const char* r1 = region + tab.a1;
    f32x4 cr[2][8];
    auto mfmas = [&](int s0, int s1) {
#pragma unroll
      for (int s = s0; s < s1; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s >> 2][s & 3], cur[2 * s + nt], acc[m][nt], 0, 0, 0);
    };
    // phase 1: k-steps of quads 0, 1 | corner reads of quads 0, 1 of the next tap
    __builtin_amdgcn_sched_barrier(0);
    if (chunk == 1) { LEAN_STAMP4(t, 0) }
    if (!last) { LEAN_LOAD8(cr[0], 0) LEAN_LOAD8(cr[1], 1) }
    mfmas(0, 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    // (hard fences between the phases: vector-ALU work scheduled in between MFMAs costs a pipe switch each time -- 4 v_pk_fma_f32 behind
    // an MFMA: +34 clocks, tools/lean_probe2.hip)
    __builtin_amdgcn_sched_barrier(0);
    if (chunk == 1) { LEAN_STAMP4(t, 1) }
    if (!last) accum2(tab, cr[0], cr[1], nxt, nxt + 4, true);
    __builtin_amdgcn_sched_barrier(0);
    if (chunk == 1) { LEAN_STAMP4(t, 2) }
    // phase 2: remaining k-steps | corner reads of the remaining quads
    if (!last) {
      LEAN_LOAD8(cr[0], 2)
      if (NQ > 3) { LEAN_LOAD8(cr[1], 3) }
    }
    mfmas(4, KSTEPS);
#pragma unroll
    for (int i = 0; i < 4 * (KSTEPS - 4); ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, 1, 1); }
    __builtin_amdgcn_sched_barrier(0);
    if (chunk == 1) { LEAN_STAMP4(t, 3) }
    if (!last) {
      accum2(tab, cr[0], cr[1], nxt + 8, nxt + 12, NQ > 3);
      slow_fix(tab, nxt, c0);
    }
    // table of the tap after next (tap t + 2; over the chunk boundary: tap t + 2 - 27 of the next chunk)
    {
      int u = t + 2;
      if (u >= T) u -= T;
      const int ti = u / 9, tj = (u - 9 * ti) / 3, tk = u - 9 * ti - 3 * tj;
      const int sl = u % 3;                            // ring slots follow the tap number modulo 3 (27 = 0 mod 3: continuous over chunks)
      const float fdn = (zbf + (float)ti) + od[sl], fhn = (ybf + (float)tj) + oh[sl], fwn = (xbf + (float)tk) + ow[sl];
      int v = u + 3;                                   // refill the slot with the tap three further on
      if (v >= T) v -= T;
      const float* np = offp0 + (long long)v * P3;
      od[sl] = np[0]; oh[sl] = np[p.P]; ow[sl] = np[2 * p.P];
      const LeanTab tn2 = lean_tab<G>(p, pvalid, ry0, rx0, fdn, fhn, fwn);
      if (last) tab1 = tn2;                            // tap 26: `tab` already is tap 0's table of the next chunk, this one is tap 1's
      else tab = tn2;
    }
  };

  float sA[16], sB[16];
  f32x4 aA[MT][2], aB[MT][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { sA[i] = 0.f; sB[i] = 0.f; }
  {
    const float fd1 = zbf + od[1], fh1 = ybf + oh[1], fw1 = (xbf + 1.f) + ow[1];      // tap 1 = (ti, tj, tk) = (0, 0, 1)
    const float* np = offp0 + 4 * P3;
    od[1] = np[0]; oh[1] = np[p.P]; ow[1] = np[2 * p.P];
    tab1 = lean_tab<G>(p, pvalid, ry0, rx0, fd1, fh1, fw1);
  }
  int chunk = 0;
#pragma unroll 1
  for (int c0 = 0; c0 < p.C; c0 += CH, ++chunk) {
    __syncthreads();                                   // every wave is done with the previous chunk's region
    lean_stage<G>(p, xb, c0, region, ry0, rx0, tid, 256);
    {                                                  // this chunk's first weight fragments
      const char* wb = reinterpret_cast<const char*>(wl) + ((long long)chunk * MT) * 2048;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        aA[m][0] = *reinterpret_cast<const f32x4*>(wb + wlane + m * 2048);
        aA[m][1] = *reinterpret_cast<const f32x4*>(wb + wlane + m * 2048 + 16);
      }
    }
    __syncthreads();
    {                                                  // tap 0 of the chunk: nothing to contract yet
      const char* r0 = region + tab.a0;
      const char* r1 = region + tab.a1;
      f32x4 cr[2][8];
      LEAN_LOAD8(cr[0], 0) LEAN_LOAD8(cr[1], 1)
      accum2(tab, cr[0], cr[1], sA, sA + 4, true);
      LEAN_LOAD8(cr[0], 2)
      if (NQ > 3) { LEAN_LOAD8(cr[1], 3) }
      accum2(tab, cr[0], cr[1], sA + 8, sA + 12, NQ > 3);
      slow_fix(tab, sA, c0);
      tab = tab1;
    }
#pragma unroll 1
    for (int t = 0; t < T - 1; t += 2) {               // taps 0 .. 25 in pairs (two register sets alternate), tap 26 below
      body(t, chunk, c0, sA, sB, aA, aB, false);
      body(t + 1, chunk, c0, sB, sA, aB, aA, false);
    }
    body(T - 1, chunk, c0, sA, sB, aA, aB, true);      // leaves tap 0's table of the next chunk in `tab`, tap 1's in `tab1`
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int vo = wave_u * 64 + nt * 32 + lane_pos32(l31);      // column n of tile nt is the voxel the sampler lane (l31, nt) owns
    const int qx = vo % G::TX, qy = (vo / G::TX) % G::TY, qz = vo / (G::TX * G::TY);
    const int gz = qz, gy = y0 + qy, gx = x0 + qx;
    // 16-byte stores: the four lanes of a quad own four x-consecutive voxels (lane_pos32), so each 4 x 4 block (rows 8 i + 4 hh + 0..3 of
    // the quad's columns) is transposed across the quad with two DPP exchange stages and lane q stores row q's four voxels at once
    // (W % 4 == 0 and x0 % 16 == 0 for every lean geometry; the store tail of 4-byte stores: conv_igemm2.hip, g2_epilogue)
    const int q = l31 & 3;
    const bool o1 = (q & 1) != 0, o2 = (q & 2) != 0;
    const bool inb = gz < p.D && gy < p.H && gx < p.W;             // uniform over a quad (gx - q is a multiple of 4, W % 4 == 0)
    const long long pos = ((long long)gz * p.H + gy) * p.W + (gx - q);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float r0 = acc[m][nt][4 * i], r1 = acc[m][nt][4 * i + 1], r2 = acc[m][nt][4 * i + 2], r3 = acc[m][nt][4 * i + 3];
        {
          const float xa = o1 ? r0 : r1, ya = o1 ? r2 : r3;
          const float xs = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0xB1, 0xf, 0xf, true));
          const float ys = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ya), 0xB1, 0xf, 0xf, true));
          if (o1) { r0 = xs; r2 = ys; } else { r1 = xs; r3 = ys; }
        }
        {
          const float xa = o2 ? r0 : r2, ya = o2 ? r1 : r3;
          const float xs = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0x4E, 0xf, 0xf, true));
          const float ys = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ya), 0x4E, 0xf, 0xf, true));
          if (o2) { r0 = xs; r1 = ys; } else { r2 = xs; r3 = ys; }
        }
        const int k = m * 32 + 8 * i + 4 * hh + q;
        if (inb && k < p.K) {
          const float bv = bias ? bias[k] : 0.f;
          const f32x4 v = {r0 + bv, r1 + bv, r2 + bv, r3 + bv};
          *reinterpret_cast<f32x4*>(out + ((long long)b * p.K + k) * p.P + pos) = v;
        }
      }
  }
}

// ---- forward, third form: the contraction on the bf16 matrix pipe (the convolutions' "x6" construction, conv_internal.h).  The fp32 MFMAs
// of dcn_lean_fwd1_kernel cost 2048 clocks per tap and wave and nothing else issues beside them; as bf16 MFMAs the same GEMM is 24 x 32 = 768
// clocks and the vector ALU stays free, so the sampler (~1300 clocks per tap with the operand split) runs in their shadow.
//   * a lane's 16 samples (fp32, one voxel) become the B operands of both 32-voxel column tiles with 8 v_permlane32_swap_b32 -- (s[i], s[8+i])
//     -> (tile 0: channels 8 hh + i of the low half's voxels, tile 1: of the high half's) -- and are split ONCE into three bf16 terms;
//   * the weights are split by the pack kernel: wl[tap][chunk][row tile][hi | mid | lo][lane][8 bf16], value i of lane (l31, hh) =
//     W[32 m + l31][chunk * CH + 8 hh + i][tap] (zero beyond K / C / CH); three 16-byte loads per tap and row tile;
//   * six MFMAs per (row tile, column tile): hi*lo, lo*hi, mid*mid, mid*hi, hi*mid, hi*hi (DPF_X9_FIRST = 3).
typedef __bf16 lean_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned lean_u32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void lean_repack_fwd6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wl, int K, int C, int MT, int nchunk) {
  const int total = 27 * nchunk * MT * 512;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63;
    int r = i >> 9;
    const int m = r % MT; r /= MT;
    const int chunk = r % nchunk;
    const int t = r / nchunk;
    const int k = 32 * m + (lane & 31), cl = 8 * (lane >> 5) + e, c = chunk * CH + cl;
    const float v = (cl < CH && k < K && c < C) ? w[((long long)k * C + c) * 27 + t] : 0.f;
    unsigned h, md, lo;
    dpf_split_pair(v, 0.f, h, md, lo);
    const long long base = ((long long)((t * nchunk + chunk) * MT + m) * 3) * 512 + lane * 8 + e;
    wl[base] = (unsigned short)(h & 0xffffu);
    wl[base + 512] = (unsigned short)(md & 0xffffu);
    wl[base + 1024] = (unsigned short)(lo & 0xffffu);
  }
}

template <class G, int MT>
__global__ __launch_bounds__(256, 2) void dcn_lean_fwd6_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                            const unsigned short* __restrict__ wl, const float* __restrict__ bias,
                                                            float* __restrict__ out, LeanP p) {
  extern __shared__ __align__(16) char smem[];
  constexpr int CH = G::CH, NQ = G::NQ, T = 27;
  static_assert(G::NV == 256, "four waves of 64 voxels");
  char* region = smem;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  int blk = lean_xcd_tile(blockIdx.x, gridDim.x);
  const int tx = blk % p.tilesX; blk /= p.tilesX;
  const int ty = blk % p.tilesY;
  const int b = blk / p.tilesY;
  const int y0 = ty * G::TY, x0 = tx * G::TX;
  const int ry0 = y0 - 1 - G::RYH, rx0 = x0 - 1 - G::RXL;
  const float* xb = x + (long long)b * p.C * p.P;
  const int vox = wave_u * 64 + (lane & 32) + lane_pos32(l31);
  const int px = vox % G::TX, py = (vox / G::TX) % G::TY, pz = vox / (G::TX * G::TY);
  const int zo = pz, yo = y0 + py, xo = x0 + px;
  const bool pvalid = zo < p.D && yo < p.H && xo < p.W;
  const long long ppos = pvalid ? ((long long)zo * p.H + yo) * p.W + xo : 0;
  const float* offp0 = offset + (long long)b * 3 * T * p.P + ppos;
  const float zbf = (float)(zo - 1), ybf = (float)(yo - 1), xbf = (float)(xo - 1);
  const long long P3 = 3 * p.P;
  const unsigned wlane = (unsigned)lane * 16u;
  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[m][nt][j] = 0.f;

  // accumulate ONE quad of a tap from its 8 corner reads into 4 sample registers (two independent packed multiply-add chains)
  auto accum1 = [&](const f32x2& w00, const f32x2& w01, const f32x2& w10, const f32x2& w11, const f32x4* c, float* sv) {
    f32x2 lo = pk_mul_lo(w00, c[0].xy), hi = pk_mul_lo(w00, c[0].zw);
    pk_fma_hi(lo, w00, c[1].xy); pk_fma_hi(hi, w00, c[1].zw);
    pk_fma_lo(lo, w01, c[2].xy); pk_fma_lo(hi, w01, c[2].zw);
    pk_fma_hi(lo, w01, c[3].xy); pk_fma_hi(hi, w01, c[3].zw);
    pk_fma_lo(lo, w10, c[4].xy); pk_fma_lo(hi, w10, c[4].zw);
    pk_fma_hi(lo, w10, c[5].xy); pk_fma_hi(hi, w10, c[5].zw);
    pk_fma_lo(lo, w11, c[6].xy); pk_fma_lo(hi, w11, c[6].zw);
    pk_fma_hi(lo, w11, c[7].xy); pk_fma_hi(hi, w11, c[7].zw);
    sv[0] = lo.x; sv[1] = lo.y; sv[2] = hi.x; sv[3] = hi.y;
  };
  // samples that leave the staged box: redone by the wave from global memory (lane = channel x corner pair), then handed to the owning lane
  auto slow_fix = [&](const LeanTab& tb, float* sv, int c0) {
    unsigned long long slow = __ballot(tb.slow);
    while (slow) {
      const int L = __builtin_ctzll(slow);
      slow &= slow - 1;
      const int sd0 = __builtin_amdgcn_readlane(tb.d0, L), sh0 = __builtin_amdgcn_readlane(tb.h0, L), sw0 = __builtin_amdgcn_readlane(tb.w0, L);
      const float sld = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb.ld), L));
      const float slh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb.lh), L));
      const float slw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb.lw), L));
      const int ch = lane & 15, jd = lane >> 5, jh = (lane >> 4) & 1;
      const int dz = sd0 + jd, hy = sh0 + jh;
      const int cgl = c0 + ch;
      const bool rowin = ch < CH && cgl < p.C && (unsigned)dz < (unsigned)p.D && (unsigned)hy < (unsigned)p.H;
      const bool in0 = rowin && (unsigned)sw0 < (unsigned)p.W, in1 = rowin && (unsigned)(sw0 + 1) < (unsigned)p.W;
      const float* xr = xb + (long long)(cgl < p.C ? cgl : 0) * p.P + ((long long)(rowin ? dz : 0) * p.H + (rowin ? hy : 0)) * p.W;
      const float v0 = in0 ? xr[sw0] : 0.f, v1 = in1 ? xr[sw0 + 1] : 0.f;
      const float wzy = (jd ? sld : 1.f - sld) * (jh ? slh : 1.f - slh);
      float part = fmaf(wzy * slw, v1, (wzy * (1.f - slw)) * v0);
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);                // lanes 0 .. 15 (and their copies) hold channel `lane & 15` of voxel L
#pragma unroll
      for (int ch2 = 0; ch2 < CH; ++ch2) {
        const float vch = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), ch2));
        if (lane == L) sv[ch2] = vch;
      }
    }
  };
  // one tap's samples (16 fp32 of the lane's voxel) -> the split bf16 B operands of both column tiles
  auto make_b = [&](float (&sv)[16], lean_u32x4 (&bh)[2], lean_u32x4 (&bm)[2], lean_u32x4 (&bl)[2]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(sv[i]), "+v"(sv[8 + i]));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        dpf_split_pair(sv[8 * nt + 2 * j], sv[8 * nt + 2 * j + 1], h, m, l);
        bh[nt][j] = h; bm[nt][j] = m; bl[nt][j] = l;
      }
  };

  // offsets ring: slot (u % 3) holds tap u's three components, fetched four taps ahead
  float od[3], oh[3], ow[3];
  LeanTab tab = lean_tab<G>(p, pvalid, ry0, rx0, zbf + offp0[0], ybf + offp0[p.P], xbf + offp0[2 * p.P]);
#pragma unroll
  for (int u = 1; u <= 3; ++u) { od[u % 3] = offp0[u * P3]; oh[u % 3] = offp0[u * P3 + p.P]; ow[u % 3] = offp0[u * P3 + 2 * p.P]; }
  LeanTab tab1;                                        // tap 1's table while tap 0 is gathered at a chunk's start

  lean_u32x4 bH[2], bM[2], bL[2];                      // B operands of the tap being contracted
  lean_u32x4 aC[MT][3], aN[MT][3];                     // its weight fragments [row tile][hi, mid, lo] and the next tap's
  float nxt[16];
  auto load_a = [&](int tn, int cn, lean_u32x4 (&a)[MT][3]) {
    const char* wb = reinterpret_cast<const char*>(wl) + ((long long)(tn * p.nchunk + cn) * MT) * 3072;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[m][c] = *reinterpret_cast<const lean_u32x4*>(wb + (m * 3 + c) * 1024 + wlane);
  };
  // sample tap `tab` into nxt (corner reads quad by quad)
  auto sample_core = [&]() {
    const char* r0 = region + tab.a0;
    const char* r1 = region + tab.a1;
    const f32x2 zy0 = pk_mul_lo(tab.wz, tab.wy), zy1 = pk_mul_hi(tab.wz, tab.wy);
    const f32x2 w00 = pk_mul_lo(zy0, tab.wx), w01 = pk_mul_hi(zy0, tab.wx), w10 = pk_mul_lo(zy1, tab.wx), w11 = pk_mul_hi(zy1, tab.wx);
    f32x4 cr[2][8];
    LEAN_LOAD8(cr[0], 0)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + 1 < NQ) { LEAN_LOAD8(cr[(q + 1) & 1], q + 1) }
      accum1(w00, w01, w10, w11, cr[q & 1], nxt + 4 * q);
    }
#pragma unroll
    for (int i = 4 * NQ; i < 16; ++i) nxt[i] = 0.f;
  };
  // the table of tap u (ring slot u % 3) and the refill of that slot with tap u + 3
  auto next_table = [&](int u) {
    const int ti = u / 9, tj = (u - 9 * ti) / 3, tk = u - 9 * ti - 3 * tj;
    const int sl = u % 3;
    const float fdn = (zbf + (float)ti) + od[sl], fhn = (ybf + (float)tj) + oh[sl], fwn = (xbf + (float)tk) + ow[sl];
    int v = u + 3;
    if (v >= T) v -= T;
    const float* np = offp0 + (long long)v * P3;
    od[sl] = np[0]; oh[sl] = np[p.P]; ow[sl] = np[2 * p.P];
    return lean_tab<G>(p, pvalid, ry0, rx0, fdn, fhn, fwn);
  };
  auto mfmas = [&]() {
    // smallest partial products first; row tiles and column tiles alternate so that consecutive MFMAs never share an accumulator
    constexpr int oa[6] = {2, 0, 1, 1, 0, 0};          // weight component   (0 = hi, 1 = mid, 2 = lo)
    constexpr int ob[6] = {0, 2, 1, 0, 1, 0};          // sample component
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const lean_u32x4 bv = ob[i] == 0 ? bH[nt] : (ob[i] == 1 ? bM[nt] : bL[nt]);
          acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(lean_bf16x8, aC[m][oa[i]]), __builtin_bit_cast(lean_bf16x8, bv), acc[m][nt], 0, 0, 0);
        }
  };

  {
    const float fd1 = zbf + od[1], fh1 = ybf + oh[1], fw1 = (xbf + 1.f) + ow[1];      // tap 1 = (ti, tj, tk) = (0, 0, 1)
    const float* np = offp0 + 4 * P3;
    od[1] = np[0]; oh[1] = np[p.P]; ow[1] = np[2 * p.P];
    tab1 = lean_tab<G>(p, pvalid, ry0, rx0, fd1, fh1, fw1);
  }
  int chunk = 0;
#pragma unroll 1
  for (int c0 = 0; c0 < p.C; c0 += CH, ++chunk) {
    __syncthreads();                                   // every wave is done with the previous chunk's region
    lean_stage<G>(p, xb, c0, region, ry0, rx0, tid, 256);
    load_a(0, chunk, aC);
    __syncthreads();
    sample_core();                                     // tap 0 of the chunk: nothing to contract yet
    slow_fix(tab, nxt, c0);
    make_b(nxt, bH, bM, bL);
    tab = tab1;
#pragma unroll 1
    for (int t = 0; t < T - 1; ++t) {
      // contract tap t (B operands / weight fragments in registers) while tap t + 1 is sampled by the vector ALU in the MFMAs' shadow: the
      // MFMAs and the corner reads / multiply-adds of the sampler form ONE basic block, interleaved by the scheduling hints
      load_a(t + 1, chunk, aN);
      __builtin_amdgcn_sched_barrier(0);
      mfmas();
      sample_core();
      int u = t + 2;
      if (u >= T) u -= T;
      const LeanTab tabn = next_table(u);                // (t = T - 2: tap 0's table of the next chunk)
      // phase A: the corner reads of the first two quads go out between the first MFMAs; phase B: one MFMA, then the multiply-adds of a
      // quad whose reads have landed, the reads of the next quad, the next table's arithmetic
#pragma unroll
      for (int i = 0; i < 4 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MT == 2 ? 2 : 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MT == 2 ? 2 : 4, 0);
      }
#pragma unroll
      for (int i = 0; i < 8 * MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MT == 2 ? 9 : 18, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, MT == 2 ? 1 : 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      slow_fix(tab, nxt, c0);
      tab = tabn;
      make_b(nxt, bH, bM, bL);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < 3; ++c) aC[m][c] = aN[m][c];
    }
    // tap 26: nothing left to sample in this chunk
    mfmas();
    tab1 = next_table(1);                              // tap 1's table of the next chunk (`tab` already is tap 0's)
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int vo = wave_u * 64 + nt * 32 + lane_pos32(l31);      // column n of tile nt is the voxel the sampler lane (l31, nt) owns
    const int qx = vo % G::TX, qy = (vo / G::TX) % G::TY, qz = vo / (G::TX * G::TY);
    const int gz = qz, gy = y0 + qy, gx = x0 + qx;
    const int q = l31 & 3;
    const bool o1 = (q & 1) != 0, o2 = (q & 2) != 0;
    const bool inb = gz < p.D && gy < p.H && gx < p.W;             // uniform over a quad (gx - q is a multiple of 4, W % 4 == 0)
    const long long pos = ((long long)gz * p.H + gy) * p.W + (gx - q);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float r0 = acc[m][nt][4 * i], r1 = acc[m][nt][4 * i + 1], r2 = acc[m][nt][4 * i + 2], r3 = acc[m][nt][4 * i + 3];
        {
          const float xa = o1 ? r0 : r1, ya = o1 ? r2 : r3;
          const float xs = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0xB1, 0xf, 0xf, true));
          const float ys = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ya), 0xB1, 0xf, 0xf, true));
          if (o1) { r0 = xs; r2 = ys; } else { r1 = xs; r3 = ys; }
        }
        {
          const float xa = o2 ? r0 : r2, ya = o2 ? r1 : r3;
          const float xs = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0x4E, 0xf, 0xf, true));
          const float ys = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, ya), 0x4E, 0xf, 0xf, true));
          if (o2) { r0 = xs; r1 = ys; } else { r2 = xs; r3 = ys; }
        }
        const int k = m * 32 + 8 * i + 4 * hh + q;
        if (inb && k < p.K) {
          const float bv = bias ? bias[k] : 0.f;
          const f32x4 v = {r0 + bv, r1 + bv, r2 + bv, r3 + bv};
          *reinterpret_cast<f32x4*>(out + ((long long)b * p.K + k) * p.P + pos) = v;
        }
      }
  }
}

// ====================================================================================================================================
// grad_offset + grad_weight (reference: deformable_col2im_coord, deform_im2col_cuda.cuh:111-190,336-405; the grad_weight GEMM,
// deform_conv_cuda.cu:220-279).  Per (chunk, tap) step:
//   gcol[c][v] = sum_k W[k][c][t] go[k][v]                      (matrix waves, v_mfma_f32_16x16x4_f32: rows = channels)
//   S[c][v]    = trilinear sample of x[c] at (v, t);  grad_offset[3t + axis][v] += sum_c gcol[c][v] dS[c][v] / d(coord)     (samplers)
//   dW[k][c][t] += sum_v go[k][v] S[c][v]                        (matrix waves)
// Eight waves: 4 samplers (a voxel per lane, all CH channels -- the lean sampler of the forward kernel plus one dot product per corner)
// and 4 matrix waves that each run BOTH products (on this chip a wave issuing fp32 MFMAs back to back starves the other waves of its
// SIMD of vector and LDS issue -- tools/lean_probe.hip -- so the two matrix roles of round 3, one wave each, were serialised anyway).
// Tiles: G[2][q][voxel][4] (gcol, quad planar: the 16x16 MFMA result of a lane IS one quad of one voxel -> one ds_write_b128; the sampler
// reads its voxel's CH values as NQ ds_read_b128) and S[2][CH][260] (channel major: the weight-gradient B operands of 4 consecutive voxels
// are one ds_read_b128).  Step i: matrix waves write G(i + 1) and contract S(i - 1); samplers read G(i), write S(i); one barrier.
constexpr int LEAN_SS = 260;          // padded row of the S tile (floats)
constexpr int LEAN_NREP = DCN_WG_NREP; // replicas of the grad_weight scratch tensor (dcn_internal.h; layout [rep][T][nchunk][64][16], folded by dcn3d.hip)

template <class G>
struct BwdLds {
  static constexpr int GT = G::NQ * G::SQ;                     // one gcol tile (bytes)
  static constexpr int ST = G::CH * LEAN_SS * 4;               // one sample tile (bytes)
  static constexpr int OFF_G = G::NQ * G::PLANE;
  static constexpr int OFF_S = OFF_G + 2 * GT;
  static constexpr int LDS = OFF_S + 2 * ST;
  static_assert(G::NV == 256, "4 sampler waves + 4 matrix waves of 64 voxels");
};

// GH = true (the default unless dpf_set_f32_matrix_path(0)): the gcol product on v_mfma_f32_16x16x32_f16 from two f16 components per operand
// (conv_internal.h) -- the weights are split and scaled by the repack kernel, the go tile once per workgroup, three MFMAs per (16 voxels,
// 32 output channels) instead of eight fp32 ones per 4; the result is scaled back exactly before it is written to the gcol tile, so the
// samplers are unchanged.  RANGE GUARD: a gcol element sums over the OUTPUT CHANNELS of one voxel, and the voxel is the MFMA's column,
// so every voxel is scaled by its own largest magnitude (a lane and its three partners hold all channels of a voxel: two shuffles) --
// gcol of a voxel is exact to fp32 relative to that voxel's output gradient, whatever its neighbours hold.
typedef _Float16 lean_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned lean_u32x4 __attribute__((ext_vector_type(4)));
template <class G, bool GH>
__global__ __launch_bounds__(512) void dcn_lean_bwd_offset_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                                  const float* __restrict__ wg /*[T][nchunk][64 lanes][16]*/,
                                                                  const float* __restrict__ go, float* __restrict__ doff, float* __restrict__ dwtmp,
                                                                  LeanP p, int det) {
  extern __shared__ __align__(16) char smem[];
  constexpr int CH = G::CH, NQ = G::NQ, T = 27;
  typedef BwdLds<G> L;
  char* region = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  int blk = lean_xcd_tile(blockIdx.x, gridDim.x);
  const int tx = blk % p.tilesX; blk /= p.tilesX;
  const int ty = blk % p.tilesY;
  const int b = blk / p.tilesY;
  const int y0 = ty * G::TY, x0 = tx * G::TX;
  const int ry0 = y0 - 1 - G::RYH, rx0 = x0 - 1 - G::RXL;
  const float* xb = x + (long long)b * p.C * p.P;
  const float* gob = go + (long long)b * p.K * p.P;
  const int NS = p.nchunk * T;
  if (wave_u < 4) {
    // ------------------------------------------------------------------------------------------------------------ samplers
    const int vox = wave_u * 64 + (lane & 32) + lane_pos32(lane & 31);
    const int px = vox % G::TX, py = (vox / G::TX) % G::TY, pz = vox / (G::TX * G::TY);
    const int zo = pz, yo = y0 + py, xo = x0 + px;
    const bool pvalid = zo < p.D && yo < p.H && xo < p.W;
    const long long ppos = pvalid ? ((long long)zo * p.H + yo) * p.W + xo : 0;
    const float* offp0 = offset + (long long)b * 3 * T * p.P + ppos;
    float* doffp0 = doff + (long long)b * 3 * T * p.P + ppos;
    const float zbf = (float)(zo - 1), ybf = (float)(yo - 1), xbf = (float)(xo - 1);
    const long long P3 = 3 * p.P;
    float od[3], oh[3], ow[3];
    LeanTab tab = lean_tab<G>(p, pvalid, ry0, rx0, zbf + offp0[0], ybf + offp0[p.P], xbf + offp0[2 * p.P]);
#pragma unroll
    for (int u = 1; u <= 3; ++u) { od[u % 3] = offp0[u * P3]; oh[u % 3] = offp0[u * P3 + p.P]; ow[u % 3] = offp0[u * P3 + 2 * p.P]; }
    lean_stage<G>(p, xb, 0, region, ry0, rx0, tid, 512);
    __syncthreads();                                   // prologue: region of chunk 0 staged, gcol(0) written
    int i = 0;
#pragma unroll 1
    for (int c0 = 0; c0 < p.C; c0 += CH) {
      if (c0 > 0) {
        lean_stage<G>(p, xb, c0, region, ry0, rx0, tid, 512);
        __syncthreads();
      }
#pragma unroll 1
      for (int g = 0; g < 9; ++g) {
        const int gn = g < 8 ? g + 1 : 0;
        const int ti = g / 3, tj = g - 3 * ti, tin = gn / 3, tjn = gn - 3 * tin;
        const float fz_same = zbf + (float)ti, fy_same = ybf + (float)tj, fz_next = zbf + (float)tin, fy_next = ybf + (float)tjn;
#pragma unroll
        for (int r = 0; r < 3; ++r, ++i) {
          const int t = 3 * g + r;
          const int sl = (r + 1) % 3;
          const float fdn = (r < 2 ? fz_same : fz_next) + od[sl], fhn = (r < 2 ? fy_same : fy_next) + oh[sl];
          const float fwn = (xbf + (float)((r + 1) % 3)) + ow[sl];
          {
            int u = t + 4;
            if (u >= T) u -= T;
            const float* np = offp0 + (long long)u * P3;
            od[sl] = np[0]; oh[sl] = np[p.P]; ow[sl] = np[2 * p.P];
          }
          const char* gt = smem + L::OFF_G + (i & 1) * L::GT + vox * 16;
          float* st = reinterpret_cast<float*>(smem + L::OFF_S + (i & 1) * L::ST);
          const char* r0 = region + tab.a0;
          const char* r1 = region + tab.a1;
          const f32x2 w00 = pk_mul_lo(pk_mul_lo(tab.wz, tab.wy), tab.wx), w01 = pk_mul_hi(pk_mul_lo(tab.wz, tab.wy), tab.wx);
          const f32x2 w10 = pk_mul_lo(pk_mul_hi(tab.wz, tab.wy), tab.wx), w11 = pk_mul_hi(pk_mul_hi(tab.wz, tab.wy), tab.wx);
          f32x4 cr[2][8], gq[2];
          LEAN_LOAD8(cr[0], 0)
          gq[0] = *reinterpret_cast<const f32x4*>(gt);
          LeanTab tabn = lean_tab<G>(p, pvalid, ry0, rx0, fdn, fhn, fwn);      // next tap's table, under the LDS latency
          f32x2 dot[8];
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            if (q + 1 < NQ) {
              LEAN_LOAD8(cr[(q + 1) & 1], q + 1)
              gq[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(gt + (q + 1) * G::SQ);
            }
            const f32x4* c = cr[q & 1];
            const f32x2 glo = gq[q & 1].xy, ghi = gq[q & 1].zw;
            // sample of the quad's 4 channels
            f32x2 lo = pk_mul_lo(w00, c[0].xy), hi = pk_mul_lo(w00, c[0].zw);
            pk_fma_hi(lo, w00, c[1].xy); pk_fma_hi(hi, w00, c[1].zw);
            pk_fma_lo(lo, w01, c[2].xy); pk_fma_lo(hi, w01, c[2].zw);
            pk_fma_hi(lo, w01, c[3].xy); pk_fma_hi(hi, w01, c[3].zw);
            pk_fma_lo(lo, w10, c[4].xy); pk_fma_lo(hi, w10, c[4].zw);
            pk_fma_hi(lo, w10, c[5].xy); pk_fma_hi(hi, w10, c[5].zw);
            pk_fma_lo(lo, w11, c[6].xy); pk_fma_lo(hi, w11, c[6].zw);
            pk_fma_hi(lo, w11, c[7].xy); pk_fma_hi(hi, w11, c[7].zw);
            st[(4 * q + 0) * LEAN_SS + vox] = lo.x; st[(4 * q + 1) * LEAN_SS + vox] = lo.y;
            st[(4 * q + 2) * LEAN_SS + vox] = hi.x; st[(4 * q + 3) * LEAN_SS + vox] = hi.y;
            // dot_j += sum over the quad's channels of gcol * corner j (two channels per lane of the packed multiply-add)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (q == 0) dot[j] = glo * c[j].xy; else dot[j] = __builtin_elementwise_fma(glo, c[j].xy, dot[j]);
              dot[j] = __builtin_elementwise_fma(ghi, c[j].zw, dot[j]);
            }
          }
          // d sample / d coord, factored: A[jd][jh] = sum_jw wx dot, E[jd][jw] = sum_jh wy dot  (dots of corners outside the volume are
          // zero along y / x because the staged image is; the depth planes carry their masks in wz / mz)
          float gd, gh, gw;
          {
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = dot[j].x + dot[j].y;            // index 4 jd + 2 jh + jw
            const float wx0 = tab.wx.x, wx1 = tab.wx.y, wy0 = tab.wy.x, wy1 = tab.wy.y, wz0 = tab.wz.x, wz1 = tab.wz.y;
            const float A00 = fmaf(wx1, d[1], wx0 * d[0]), A01 = fmaf(wx1, d[3], wx0 * d[2]);
            const float A10 = fmaf(wx1, d[5], wx0 * d[4]), A11 = fmaf(wx1, d[7], wx0 * d[6]);
            const float E00 = fmaf(wy1, d[2], wy0 * d[0]), E01 = fmaf(wy1, d[3], wy0 * d[1]);
            const float E10 = fmaf(wy1, d[6], wy0 * d[4]), E11 = fmaf(wy1, d[7], wy0 * d[5]);
            const float B0 = fmaf(wy1, A01, wy0 * A00), B1 = fmaf(wy1, A11, wy0 * A10);
            gd = fmaf(tab.mz.y, B1, -tab.mz.x * B0);
            gh = fmaf(wz1, A11 - A10, wz0 * (A01 - A00));
            gw = fmaf(wz1, E11 - E10, wz0 * (E01 - E00));
          }
          // ---- samples that leave the staged box: the wave redoes them from global memory (lane = channel x corner pair (jd, jh))
          unsigned long long slow = __ballot(tab.slow);
          while (slow) {
            const int Lq = __builtin_ctzll(slow);
            slow &= slow - 1;
            const int sd0 = __builtin_amdgcn_readlane(tab.d0, Lq), sh0 = __builtin_amdgcn_readlane(tab.h0, Lq), sw0 = __builtin_amdgcn_readlane(tab.w0, Lq);
            const float sld = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.ld), Lq));
            const float slh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.lh), Lq));
            const float slw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tab.lw), Lq));
            const int svox = __builtin_amdgcn_readlane(vox, Lq);
            const int ch = lane & 15, jd = lane >> 5, jh = (lane >> 4) & 1;
            const int dz = sd0 + jd, hy = sh0 + jh;
            const int cgl = c0 + ch;
            const bool rowin = ch < CH && cgl < p.C && (unsigned)dz < (unsigned)p.D && (unsigned)hy < (unsigned)p.H;
            const bool in0 = rowin && (unsigned)sw0 < (unsigned)p.W, in1 = rowin && (unsigned)(sw0 + 1) < (unsigned)p.W;
            const float* xr = xb + (long long)(cgl < p.C ? cgl : 0) * p.P + ((long long)(rowin ? dz : 0) * p.H + (rowin ? hy : 0)) * p.W;
            const float v0 = in0 ? xr[sw0] : 0.f, v1 = in1 ? xr[sw0 + 1] : 0.f;
            const float gch = ch < CH ? *reinterpret_cast<const float*>(smem + L::OFF_G + (i & 1) * L::GT + (ch >> 2) * G::SQ + svox * 16 + (ch & 3) * 4) : 0.f;
            const float wzj = jd ? sld : 1.f - sld, wyj = jh ? slh : 1.f - slh;
            float part = (wzj * wyj) * fmaf(slw, v1, (1.f - slw) * v0);
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (lane < CH) st[lane * LEAN_SS + svox] = part;
            float e0 = gch * v0, e1 = gch * v1;          // this channel's share of dot(jd, jh, 0 / 1); summed over the 16 channels of the row
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { e0 += __shfl_xor(e0, m, 64); e1 += __shfl_xor(e1, m, 64); }
            // every lane of row (jd, jh) now holds that corner pair's two dots; the three derivatives are sums over the four rows
            const float ex = fmaf(slw, e1, (1.f - slw) * e0);                     // sum_jw wx dot
            float sgd = (jd ? 1.f : -1.f) * wyj * ex;                              // d/dz: sign of the plane, other factors wy wx
            float sgh = (jh ? 1.f : -1.f) * wzj * ex;
            float sgw = wzj * wyj * (e1 - e0);
            sgd += __shfl_xor(sgd, 16, 64); sgd += __shfl_xor(sgd, 32, 64);
            sgh += __shfl_xor(sgh, 16, 64); sgh += __shfl_xor(sgh, 32, 64);
            sgw += __shfl_xor(sgw, 16, 64); sgw += __shfl_xor(sgw, 32, 64);
            if (lane == Lq) { gd = sgd; gh = sgh; gw = sgw; }
          }
          if (pvalid) {
            float* dq = doffp0 + (long long)(3 * t) * p.P;
            if (c0 == 0) { dq[0] = gd; dq[p.P] = gh; dq[2 * p.P] = gw; }
            else { atomicAdd(dq, gd); atomicAdd(dq + p.P, gh); atomicAdd(dq + 2 * p.P, gw); }
          }
          tab = tabn;
          __syncthreads();                             // step barrier
        }
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------------------ matrix waves
    const int mw = wave_u - 4, l15 = lane & 15, lg = lane >> 4;
    // gcol B fragments: go[k = 4 ks + lg][voxel] for this wave's 4 sub-tiles of 16 voxels (GH: k = 32 mf + 8 lg + i, packed f16 components)
    float bfrag[GH ? 1 : 4][16];
    lean_u32x4 bq[GH ? 4 : 1][2][2];                    // [sub-tile][k half][hi | lo]
    int Egv[GH ? 4 : 1], Ew = DPF_H3_EMIN;               // GH: exponent of this lane's voxel of sub-tile s4; of the weight tensor
    if constexpr (GH) {
      Ew = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(wg)[27 * p.nchunk * 1024]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int vox = mw * 64 + s4 * 16 + l15;
        const int px = vox % G::TX, py = (vox / G::TX) % G::TY, pz = vox / (G::TX * G::TY);
        const bool ok = pz < p.D && y0 + py < p.H && x0 + px < p.W;
        const long long gpos = ok ? ((long long)pz * p.H + y0 + py) * p.W + x0 + px : 0;
        float gv[16];
        float mxg = 0.f;
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 32 * mf + 8 * lg + i;
            gv[8 * mf + i] = (ok && k < p.K) ? gob[(long long)k * p.P + gpos] : 0.f;
            mxg = __builtin_fmaxf(mxg, __builtin_fabsf(gv[8 * mf + i]));
          }
        mxg = __builtin_fmaxf(mxg, __shfl_xor(mxg, 16, 64));             // the voxel's other output channels sit in lanes l15 + 16 lg'
        mxg = __builtin_fmaxf(mxg, __shfl_xor(mxg, 32, 64));
        int e = (int)(__builtin_bit_cast(unsigned, mxg) >> 23);
        e = e < DPF_H3_EMIN ? DPF_H3_EMIN : (e > 254 ? 254 : e);
        Egv[s4] = e;
        const float scg = dpf_h3_scale(e);
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned h, l;
            dpf_split_pair_h(gv[8 * mf + 2 * i] * scg, gv[8 * mf + 2 * i + 1] * scg, h, l);
            bq[s4][mf][0][i] = h; bq[s4][mf][1][i] = l;
          }
      }
    } else {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int vox = mw * 64 + s4 * 16 + l15;
        const int px = vox % G::TX, py = (vox / G::TX) % G::TY, pz = vox / (G::TX * G::TY);
        const bool ok = pz < p.D && y0 + py < p.H && x0 + px < p.W;
        const long long gpos = ok ? ((long long)pz * p.H + y0 + py) * p.W + x0 + px : 0;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int k = 4 * ks + lg;
          bfrag[s4][ks] = (ok && k < p.K) ? gob[(long long)k * p.P + gpos] : 0.f;
        }
      }
    }
    // grad_weight A fragments: go[k = 16 mw + l15][voxel 16 q + 4 lg + u] for k-step 4 q + u
    float wfrag[64];
    {
      const int kk = 16 * mw + l15;
#pragma unroll
      for (int ks = 0; ks < 64; ++ks) {
        const int vox = 16 * (ks >> 2) + 4 * lg + (ks & 3);
        const int px = vox % G::TX, py = (vox / G::TX) % G::TY, pz = vox / (G::TX * G::TY);
        const bool ok = kk < p.K && pz < p.D && y0 + py < p.H && x0 + px < p.W;
        wfrag[ks] = ok ? gob[(long long)kk * p.P + ((long long)pz * p.H + y0 + py) * p.W + x0 + px] : 0.f;
      }
    }
    float* rep = det ? dwtmp : dwtmp + (long long)(blockIdx.x % LEAN_NREP) * T * p.nchunk * 64 * 16;
    long long* rep_shadow = det ? reinterpret_cast<long long*>(dwtmp) : nullptr;
    const int brow = (l15 < CH ? l15 : CH - 1) * LEAN_SS + 4 * lg;
    const unsigned wlane = (unsigned)lane * 64u;
    auto gcol = [&](int j) {                           // gcol(j) -> G[j & 1]
      const int chunk = j / T, t = j - chunk * T;
      const char* wb = reinterpret_cast<const char*>(wg) + (long long)(t * p.nchunk + chunk) * 4096;
      char* dst = smem + L::OFF_G + (j & 1) * L::GT + lg * G::SQ + (mw * 64 + l15) * 16;
      f32x4 a4[4];
      lean_u32x4 aq[2][2];                               // GH: [k half][hi | lo], 1 KB per (half, component)
      if constexpr (GH) {
#pragma unroll
        for (int u = 0; u < 4; ++u) aq[u >> 1][u & 1] = *reinterpret_cast<const lean_u32x4*>(wb + u * 1024 + lane * 16);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] = *reinterpret_cast<const f32x4*>(wb + wlane + 16 * u);
      }
#pragma unroll
      for (int sp2 = 0; sp2 < 2; ++sp2) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (GH) {
          constexpr int ca[3] = {1, 0, 0}, cb[3] = {0, 1, 0};          // lo*hi, hi*lo, hi*hi
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
              for (int u = 0; u < 2; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lean_f16x8, aq[mf][ca[i]]),
                                                                 __builtin_bit_cast(lean_f16x8, bq[2 * sp2 + u][mf][cb[i]]), acc[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[u][r] = __builtin_ldexpf(acc[u][r], Egv[2 * sp2 + u] + Ew - 282);
        } else {
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[ks >> 2][ks & 3], bfrag[2 * sp2 + u][ks], acc[u], 0, 0, 0);
        }
        if (lg < NQ) {
          *reinterpret_cast<f32x4*>(dst + (2 * sp2) * 256) = acc[0];            // D rows 4 lg .. 4 lg + 3 = quad lg of voxel (col)
          *reinterpret_cast<f32x4*>(dst + (2 * sp2 + 1) * 256) = acc[1];
        }
      }
    };
    gcol(0);
    lean_stage<G>(p, xb, 0, region, ry0, rx0, tid, 512);
    __syncthreads();                                   // prologue barrier
#pragma unroll 1
    for (int i = 0; i <= NS; ++i) {
      if (i + 1 < NS) gcol(i + 1);
      if (i >= 1) {                                    // contract S(i - 1)
        const int j = i - 1, cs = j / T, ts = j - cs * T;
        const float* src = reinterpret_cast<const float*>(smem + L::OFF_S + (j & 1) * L::ST) + brow;
        f32x4 wacc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(src + 16 * qq);
          wacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 0], bv.x, wacc[0], 0, 0, 0);
          wacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 1], bv.y, wacc[1], 0, 0, 0);
          wacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 2], bv.z, wacc[2], 0, 0, 0);
          wacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[4 * qq + 3], bv.w, wacc[3], 0, 0, 0);
        }
        if (l15 < CH && cs * CH + l15 < p.C) {
          float* dst = rep + ((long long)(ts * p.nchunk + cs) * 64 + 16 * mw + 4 * lg) * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * mw + 4 * lg + r < p.K) dcn_acc_add(rep, rep_shadow, &dst[r * 16], (wacc[0][r] + wacc[1][r]) + (wacc[2][r] + wacc[3][r]));
        }
      }
      if (i < NS) {
        __syncthreads();                               // barrier of step i
        if (i + 1 < NS && (i + 1) % T == 0) {          // step i was the last tap of its chunk: re-stage the region
          lean_stage<G>(p, xb, ((i + 1) / T) * CH, region, ry0, rx0, tid, 512);
          __syncthreads();
        }
      }
    }
  }
}

// gcol A fragments: wg[tap][chunk][lane][16] with entry ks of lane (l15, lg) = W[k = 4 ks + lg][c = chunk * CH + l15][tap] (zero beyond K / C / CH)
__global__ void lean_repack_gcol_kernel(const float* __restrict__ w, float* __restrict__ wg, int K, int C, int CH, int nchunk) {
  const int total = 27 * nchunk * 1024;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ks = i & 15, lane = (i >> 4) & 63;
    const int r = i >> 10;
    const int chunk = r % nchunk, t = r / nchunk;
    const int k = 4 * ks + (lane >> 4), l15 = lane & 15, c = chunk * CH + l15;
    wg[i] = (k < K && l15 < CH && c < C) ? w[((long long)k * C + c) * 27 + t] : 0.f;
  }
}

// GH: wgh[tap][chunk][k half mf][hi | lo][lane][8 halves]: value i of lane (l15, lg) = component of W[k = 32 mf + 8 lg + i][c = chunk * CH + l15][tap]
// scaled by 2^(141 - E), E = the largest exponent of the weight tensor (every workgroup finds it; workgroup 0 stores it behind the fragments)
__global__ void lean_repack_gcol_h_kernel(const float* __restrict__ w, unsigned short* __restrict__ wgh, int K, int C, int CH, int nchunk) {
  __shared__ int s_e[16];
  float m = 0.f;
  const int nw = K * C * 27;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) m = __builtin_fmaxf(m, __builtin_fabsf(w[i]));
  const int e = dpf_wave_max_exp(__builtin_bit_cast(unsigned, m));
  if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6] = e;
  __syncthreads();
  int E = DPF_H3_EMIN;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) E = s_e[i] > E ? s_e[i] : E;
  E = E > 254 ? 254 : E;
  const float sc = dpf_h3_scale(E);
  const int total = 27 * nchunk * 2048;
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(wgh)[total / 2] = E;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int idx = i & 7, lane = (i >> 3) & 63, comp = (i >> 9) & 1, mf = (i >> 10) & 1;
    const int r = i >> 11;
    const int chunk = r % nchunk, t = r / nchunk;
    const int k = 32 * mf + 8 * (lane >> 4) + idx, l15 = lane & 15, c = chunk * CH + l15;
    const float v = (k < K && l15 < CH && c < C) ? w[((long long)k * C + c) * 27 + t] * sc : 0.f;
    unsigned h, l;
    dpf_split_pair_h(v, 0.f, h, l);
    wgh[i] = (unsigned short)((comp ? l : h) & 0xffffu);
  }
}

template <typename F>
int lean_set_lds(F f, size_t lds) {
  if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DPF_ERR_LAUNCH;
  return DPF_OK;
}

template <class G>
int lean_launch_fwd1(const float* x, const float* offset, const float* weight, const float* bias, float* out, float* ws, LeanP p, hipStream_t st) {
  p.tilesY = dpf_div_up(p.H, G::TY);
  p.tilesX = dpf_div_up(p.W, G::TX);
  const long long blocks = (long long)p.B * p.tilesY * p.tilesX;
  if (blocks >= 0x7fffffffLL) return DPF_ERR_UNSUPPORTED;
  const int MT = p.KT / 32;
  constexpr int LDS = G::NQ * G::PLANE;
  hipLaunchKernelGGL((lean_repack_fwd1_kernel<G::CH>), dim3(dpf_ew_grid(27LL * p.nchunk * MT * 512)), dim3(256), 0, st, weight, ws, p.K, p.C, MT, p.nchunk);
  const dim3 grid((unsigned)blocks), block(256);
  if (MT == 1) {
    if (lean_set_lds(dcn_lean_fwd1_kernel<G, 1>, LDS) != DPF_OK) return DPF_ERR_LAUNCH;
    hipLaunchKernelGGL((dcn_lean_fwd1_kernel<G, 1>), grid, block, LDS, st, x, offset, ws, bias, out, p);
  } else {
    if (lean_set_lds(dcn_lean_fwd1_kernel<G, 2>, LDS) != DPF_OK) return DPF_ERR_LAUNCH;
    hipLaunchKernelGGL((dcn_lean_fwd1_kernel<G, 2>), grid, block, LDS, st, x, offset, ws, bias, out, p);
  }
  return dpf_check_launch();
}

template <class G>
int lean_launch_fwd6(const float* x, const float* offset, const float* weight, const float* bias, float* out, float* ws, LeanP p, hipStream_t st) {
  p.tilesY = dpf_div_up(p.H, G::TY);
  p.tilesX = dpf_div_up(p.W, G::TX);
  const long long blocks = (long long)p.B * p.tilesY * p.tilesX;
  if (blocks >= 0x7fffffffLL) return DPF_ERR_UNSUPPORTED;
  const int MT = p.KT / 32;
  constexpr int LDS = G::NQ * G::PLANE;
  unsigned short* wl = reinterpret_cast<unsigned short*>(ws);
  hipLaunchKernelGGL((lean_repack_fwd6_kernel<G::CH>), dim3(dpf_ew_grid(27LL * p.nchunk * MT * 512)), dim3(256), 0, st, weight, wl, p.K, p.C, MT, p.nchunk);
  const dim3 grid((unsigned)blocks), block(256);
  if (MT == 1) {
    if (lean_set_lds(dcn_lean_fwd6_kernel<G, 1>, LDS) != DPF_OK) return DPF_ERR_LAUNCH;
    hipLaunchKernelGGL((dcn_lean_fwd6_kernel<G, 1>), grid, block, LDS, st, x, offset, wl, bias, out, p);
  } else {
    if (lean_set_lds(dcn_lean_fwd6_kernel<G, 2>, LDS) != DPF_OK) return DPF_ERR_LAUNCH;
    hipLaunchKernelGGL((dcn_lean_fwd6_kernel<G, 2>), grid, block, LDS, st, x, offset, wl, bias, out, p);
  }
  return dpf_check_launch();
}

//                  CH TY  TX RYH RXL RXR      voxels  region cells    LDS
typedef Geo<16, 4, 16, 3, 3, 3> G16c;   //   256     4 x 12 x 24     73 728   forward: region only, two workgroups per CU
typedef Geo<12, 4, 16, 5, 7, 4> G12c;   //   256     4 x 16 x 32     98 304   forward, wider x halo: one workgroup per CU (DPF_DCN_LEAN_WIDE12=1)
typedef Geo<12, 4, 16, 5, 3, 3> G12d;   //   256     4 x 16 x 24     73 728   forward: region only, two workgroups per CU

typedef Geo<16, 4, 16, 4, 3, 3> B16;    //   256     4 x 14 x 24    152 064   backward tiles: region + 2 gcol tiles + 2 sample tiles
typedef Geo<12, 4, 16, 5, 7, 4> B12;    //   256     4 x 16 x 32    147 840

template <class G>
int lean_launch_bwd_offset(const float* x, const float* offset, const float* weight, const float* go, float* doff, float* dwtmp, float* ws, LeanP p,
                           hipStream_t st, int det) {
  p.tilesY = dpf_div_up(p.H, G::TY);
  p.tilesX = dpf_div_up(p.W, G::TX);
  const long long blocks = (long long)p.B * p.tilesY * p.tilesX;
  if (blocks >= 0x7fffffffLL) return DPF_ERR_UNSUPPORTED;
  static const int gh_env = getenv("DPF_DCN_GCOL16") ? atoi(getenv("DPF_DCN_GCOL16")) : 1;
  if (gh_env && dpf_conv_f32_x9() == 2) {
    hipLaunchKernelGGL(lean_repack_gcol_h_kernel, dim3(16), dim3(1024), 0, st, weight, reinterpret_cast<unsigned short*>(ws), p.K, p.C, G::CH, p.nchunk);
    if (lean_set_lds(dcn_lean_bwd_offset_kernel<G, true>, BwdLds<G>::LDS + 64) != DPF_OK) return DPF_ERR_LAUNCH;
    hipLaunchKernelGGL((dcn_lean_bwd_offset_kernel<G, true>), dim3((unsigned)blocks), dim3(512), BwdLds<G>::LDS + 64, st, x, offset, ws, go, doff, dwtmp, p, det);
    return dpf_check_launch();
  }
  hipLaunchKernelGGL(lean_repack_gcol_kernel, dim3(dpf_ew_grid(27LL * p.nchunk * 1024)), dim3(256), 0, st, weight, ws, p.K, p.C, G::CH, p.nchunk);
  if (lean_set_lds(dcn_lean_bwd_offset_kernel<G, false>, BwdLds<G>::LDS) != DPF_OK) return DPF_ERR_LAUNCH;
  hipLaunchKernelGGL((dcn_lean_bwd_offset_kernel<G, false>), dim3((unsigned)blocks), dim3(512), BwdLds<G>::LDS, st, x, offset, ws, go, doff, dwtmp, p, det);
  return dpf_check_launch();
}

}  // namespace

#ifdef DPF_STAMPS
extern "C" int dpf_debug_lean_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lean_stamps), sizeof(unsigned long long) * 16 * 128 * 4) == hipSuccess ? 0 : -1;
}
#endif

// chunk width: 12 where it pads the channel count less than 16 does (35 -> 36 instead of 48)
int dcn_lean_chunk(int C) { return ((C + 11) / 12 * 12 < (C + 15) / 16 * 16) ? 12 : 16; }

long long dcn_lean_workspace_floats(int C, int K) {
  const int CH = dcn_lean_chunk(C);
  const long long nchunk = (C + CH - 1) / CH;
  const long long fwd = 27LL * nchunk * ((K + 31) / 32) * 768;     // lean_repack_fwd6_kernel (3 x 512 bf16; lean_repack_fwd1_kernel: 512 floats)
  const long long bwd = 27LL * nchunk * 1024 + 16;                 // lean_repack_gcol_kernel / lean_repack_gcol_h_kernel (+ its exponent); independent of K
  return fwd > bwd ? fwd : bwd;
}

int dcn_lean_forward(const float* x, const float* offset, const float* weight, const float* bias, float* out, float* ws, int B, int C, int D, int H,
                     int W, int K, hipStream_t st) {
  static const int lean_env = getenv("DPF_DCN_LEAN") ? atoi(getenv("DPF_DCN_LEAN")) : 1;
  if (!lean_env || D > 4 || D < 1 || (W & 3) || K > 64 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return DPF_ERR_UNSUPPORTED;
  if ((long long)D * H * W >= 0x7fffffffLL / 4) return DPF_ERR_UNSUPPORTED;
  const int CH = dcn_lean_chunk(C);
  LeanP p{};
  p.B = B; p.C = C; p.K = K; p.D = D; p.H = H; p.W = W;
  p.Cpad = (C + CH - 1) / CH * CH;
  p.nchunk = p.Cpad / CH;
  p.KT = 32 * ((K + 31) / 32);
  p.P = (long long)D * H * W;
  // fp32 products: six bf16 partial products on the bf16 matrix pipe (default), or v_mfma_f32_32x32x2_f32 (dpf_set_f32_matrix_path(0), DPF_DCN_FWD6=0)
  static const int fwd6_env = getenv("DPF_DCN_FWD6") ? atoi(getenv("DPF_DCN_FWD6")) : 1;
  if (fwd6_env && dpf_conv_f32_x9()) {
    if (CH == 16) return lean_launch_fwd6<G16c>(x, offset, weight, bias, out, ws, p, st);
    return lean_launch_fwd6<G12d>(x, offset, weight, bias, out, ws, p, st);
  }
  if (CH == 16) return lean_launch_fwd1<G16c>(x, offset, weight, bias, out, ws, p, st);
  static const int wide12 = getenv("DPF_DCN_LEAN_WIDE12") ? atoi(getenv("DPF_DCN_LEAN_WIDE12")) : 0;   // 1: wider x halo, one workgroup per CU (3.7 vs 2.6 ms)
  if (wide12) return lean_launch_fwd1<G12c>(x, offset, weight, bias, out, ws, p, st);
  return lean_launch_fwd1<G12d>(x, offset, weight, bias, out, ws, p, st);
}

// grad_offset (fully written) + grad_weight partials into dwtmp[LEAN_NREP = 8][27][nchunk][64][16] (zero-initialised by the caller, folded by
// dcn3d.hip's dcn_wgrad_fold_kernel with chunk width dcn_lean_chunk(C)).  ws: >= 27 * nchunk * 1024 floats.
int dcn_lean_bwd_offset(const float* x, const float* offset, const float* weight, const float* go, float* doff, float* dwtmp, float* ws, int B, int C,
                        int D, int H, int W, int K, hipStream_t st, int det) {
  static const int lean_env = getenv("DPF_DCN_LEAN") ? atoi(getenv("DPF_DCN_LEAN")) : 1;
  if (!lean_env || (lean_env & 4) || D > 4 || D < 1 || (W & 3) || K > 64 || (reinterpret_cast<uintptr_t>(x) & 15)) return DPF_ERR_UNSUPPORTED;
  if ((long long)D * H * W >= 0x7fffffffLL / 4) return DPF_ERR_UNSUPPORTED;
  const int CH = dcn_lean_chunk(C);
  LeanP p{};
  p.B = B; p.C = C; p.K = K; p.D = D; p.H = H; p.W = W;
  p.Cpad = (C + CH - 1) / CH * CH;
  p.nchunk = p.Cpad / CH;
  p.KT = 32 * ((K + 31) / 32);
  p.P = (long long)D * H * W;
  if (CH == 16) return lean_launch_bwd_offset<B16>(x, offset, weight, go, doff, dwtmp, ws, p, st, det);
  return lean_launch_bwd_offset<B12>(x, offset, weight, go, doff, dwtmp, ws, p, st, det);
}
