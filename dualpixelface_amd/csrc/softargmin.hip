// Disparity head: trilinear x4 upsampling (align_corners) of the [B,1,D,h,w] cost logits, softmax over the
// 4*D hypotheses and the soft-argmin expectation, fused into one HBM-bound pass (reads D*h*w, writes H*W
// (+ 4*D*H*W when the probability volume is requested)).
//
// Replaces F.interpolate(mode='trilinear', align_corners=True) + squeeze + F.softmax(dim=1) + sum(p*disp)
// (reference: src/model/stereodpnet/modules.py:327-334 and :341-362).
// Backward scatters d(logit) into the low-resolution volume through an LDS-privatised tile (one LDS atomic per
// contribution, one global atomic per touched low-res cell and tile).
#include "dpf_common.h"
#include <type_traits>

namespace {

constexpr int MAXD = 8;     // low-res depth levels
constexpr int MAXL = 32;    // upsampled hypotheses
constexpr int TY = 8, TX = 32;

struct HeadP {
  int B, D, h, w, L, H, W;
  long long pbs;        // batch stride of the probability output (0: L * H * W, dense)
  float off;            // 0: align_corners=True (ratio (in-1)/(out-1)); 0.5: align_corners=False (ratio in/out, half-pixel centres)
  float disp[MAXL];
};

__device__ __forceinline__ float head_ratio(int in, int out, float off) {
  return off > 0.f ? (float)in / (float)out : (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f);
}

__device__ __forceinline__ void ac_src(int dst, float ratio, int in, int& i0, int& i1, float& lam, float off = 0.f) {
  float src = ratio * ((float)dst + off) - off;       // ATen area_pixel_compute_source_index
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = src - (float)i0;
}

__device__ __forceinline__ void pixel_probs(const float* __restrict__ k, const HeadP& p, int b, int Y, int X, float rd, float ry, float rx,
                                            float* prob, float& pred, int& y0, int& y1, int& x0, int& x1, float& ly, float& lx) {
  ac_src(Y, ry, p.h, y0, y1, ly, p.off);
  ac_src(X, rx, p.w, x0, x1, lx, p.off);
  const float hy = 1.f - ly, hx = 1.f - lx;
  float bl[MAXD];
  const float* kb = k + (long long)b * p.D * p.h * p.w;
#pragma unroll
  for (int d = 0; d < MAXD; ++d) {
    if (d < p.D) {
      const float* q = kb + (long long)d * p.h * p.w;
      bl[d] = hy * (hx * q[y0 * p.w + x0] + lx * q[y0 * p.w + x1]) + ly * (hx * q[y1 * p.w + x0] + lx * q[y1 * p.w + x1]);
    } else {
      bl[d] = 0.f;
    }
  }
  float mx = -3.4e38f;
#pragma unroll
  for (int l = 0; l < MAXL; ++l) {
    if (l < p.L) {
      int d0, d1;
      float ld;
      ac_src(l, rd, p.D, d0, d1, ld, p.off);
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int d = 0; d < MAXD; ++d) {   // register-resident select (no runtime-indexed array)
        v0 = (d == d0) ? bl[d] : v0;
        v1 = (d == d1) ? bl[d] : v1;
      }
      prob[l] = (1.f - ld) * v0 + ld * v1;
      mx = fmaxf(mx, prob[l]);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int l = 0; l < MAXL; ++l)
    if (l < p.L) {
      prob[l] = __expf(prob[l] - mx);   // v_exp_f32 path (2 ulp): the softmax is checked to 1e-5
      sum += prob[l];
    }
  pred = 0.f;
#pragma unroll
  for (int l = 0; l < MAXL; ++l)
    if (l < p.L) {
      prob[l] = prob[l] / sum;
      pred += prob[l] * p.disp[l];
    }
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ k, float* __restrict__ pred, float* __restrict__ prob, HeadP p) {
  const float rd = head_ratio(p.D, p.L, p.off), ry = head_ratio(p.h, p.H, p.off), rx = head_ratio(p.w, p.W, p.off);
  const long long total = (long long)p.B * p.H * p.W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % p.W);
    const int Y = (int)((i / p.W) % p.H);
    const int b = (int)(i / ((long long)p.W * p.H));
    float pr[MAXL];
    float e;
    int y0, y1, x0, x1;
    float ly, lx;
    pixel_probs(k, p, b, Y, X, rd, ry, rx, pr, e, y0, y1, x0, x1, ly, lx);
    pred[i] = e;
    if (prob) {
      float* q = prob + (long long)b * (p.pbs ? p.pbs : (long long)p.L * p.H * p.W) + (long long)Y * p.W + X;
#pragma unroll
      for (int l = 0; l < MAXL; ++l)
        if (l < p.L) q[(long long)l * p.H * p.W] = pr[l];
    }
  }
}

// The shipped geometry (8 cost levels -> 32 hypotheses): with D and L compile-time constants the level interpolation
// (d0, d1, lambda of every hypothesis) folds to constants, i.e. 32 FMAs on registers instead of 512 register-select instructions
__global__ __launch_bounds__(256) void head_fwd_8x32_kernel(const float* __restrict__ k, float* __restrict__ pred, float* __restrict__ prob, HeadP p) {
  constexpr int D = 8, L = 32;
  constexpr float rd = (float)(D - 1) / (float)(L - 1);
  const float ry = p.H > 1 ? (float)(p.h - 1) / (float)(p.H - 1) : 0.f;
  const float rx = p.W > 1 ? (float)(p.w - 1) / (float)(p.W - 1) : 0.f;
  const long long total = (long long)p.B * p.H * p.W;
  const long long plane = (long long)p.H * p.W, lplane = (long long)p.h * p.w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % p.W);
    const int Y = (int)((i / p.W) % p.H);
    const int b = (int)(i / plane);
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(Y, ry, p.h, y0, y1, ly);
    ac_src(X, rx, p.w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* kb = k + (long long)b * D * lplane;
    float bl[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float* q = kb + (long long)d * lplane;
      bl[d] = hy * (hx * q[y0 * p.w + x0] + lx * q[y0 * p.w + x1]) + ly * (hx * q[y1 * p.w + x0] + lx * q[y1 * p.w + x1]);
    }
    float pr[L];
    float mx = -3.4e38f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float src = rd * (float)l;                      // same expression as ac_src: folded at compile time
      int d0 = (int)src;
      if (d0 > D - 1) d0 = D - 1;
      const int d1 = d0 + (d0 < D - 1 ? 1 : 0);
      const float ld = src - (float)d0;
      pr[l] = (1.f - ld) * bl[d0] + ld * bl[d1];
      mx = fmaxf(mx, pr[l]);
    }
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      pr[l] = __expf(pr[l] - mx);
      sum += pr[l];
    }
    float e = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      pr[l] = pr[l] / sum;
      e += pr[l] * p.disp[l];
    }
    pred[i] = e;
    if (prob) {
      float* q = prob + (long long)b * (p.pbs ? p.pbs : L * plane) + (long long)Y * p.W + X;
#pragma unroll
      for (int l = 0; l < L; ++l) q[(long long)l * plane] = pr[l];
    }
  }
}

// DET (deterministic mode, dpf_common.h): (i) the LDS cells are 64-bit integers -- contributions are converted with a scale derived from the
// workgroup's largest gradient (|cell sum| < 2^60, resolution 2^-50 of that maximum), integer adds commute; (ii) the launch is split into
// Py x Px phases of tiles whose low-resolution footprints are disjoint (host: softargmin_bwd), and a workgroup outside the phase (py, px)
// exits: within a launch every dk cell receives at most one atomic add, and the phases run in stream order.
template <bool DET>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ k, const float* __restrict__ gpred, float* __restrict__ dk, HeadP p,
                                                       int Py, int Px, int py, int px) {
  // fp64 cells: on MI355X ds_add_f32 sustains 0.33 lanes/clk/CU, ds_add_f64 3.1 (profiles/r01_lds_atomic_microbench.txt)
  typedef typename std::conditional<DET, long long, double>::type cell_t;
  __shared__ cell_t tile[MAXD][TY + 2][TX + 2];
  __shared__ float s_bl[MAXD][256];   // per-thread bilinear values / gradients, thread index fastest (conflict-free)
  __shared__ float s_db[MAXD][256];
  __shared__ float s_max[4];
  const float rd = head_ratio(p.D, p.L, p.off), ry = head_ratio(p.h, p.H, p.off), rx = head_ratio(p.w, p.W, p.off);
  const int tilesX = (p.W + TX - 1) / TX, tilesY = (p.H + TY - 1) / TY;
  int bb = blockIdx.x;
  const int tx = bb % tilesX; bb /= tilesX;
  const int ty = bb % tilesY;
  const int b = bb / tilesY;
  if (DET && (ty % Py != py || tx % Px != px)) return;
  const int Y0 = ty * TY, X0 = tx * TX;
  int ybase, xbase, t1;
  float tl;
  ac_src(Y0, ry, p.h, ybase, t1, tl, p.off);
  ac_src(X0, rx, p.w, xbase, t1, tl, p.off);
  cell_t* flat = &tile[0][0][0];
  for (int i = threadIdx.x; i < MAXD * (TY + 2) * (TX + 2); i += 256) flat[i] = 0;
  __syncthreads();
  const int tid = threadIdx.x;
  const int Y = Y0 + (tid >> 5), X = X0 + (tid & 31);
  const bool active = Y < p.H && X < p.W;
  int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
  float ly = 0.f, lx = 0.f, vmax = 0.f;
  if (active) {
    ac_src(Y, ry, p.h, y0, y1, ly, p.off);
    ac_src(X, rx, p.w, x0, x1, lx, p.off);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* kb = k + (long long)b * p.D * p.h * p.w;
    for (int d = 0; d < p.D; ++d) {
      const float* q = kb + (long long)d * p.h * p.w;
      s_bl[d][tid] = hy * (hx * q[y0 * p.w + x0] + lx * q[y0 * p.w + x1]) + ly * (hx * q[y1 * p.w + x0] + lx * q[y1 * p.w + x1]);
      s_db[d][tid] = 0.f;
    }
    float mx = -3.4e38f;
    for (int l = 0; l < p.L; ++l) {
      int d0, d1;
      float ld;
      ac_src(l, rd, p.D, d0, d1, ld, p.off);
      mx = fmaxf(mx, (1.f - ld) * s_bl[d0][tid] + ld * s_bl[d1][tid]);
    }
    float sum = 0.f, num = 0.f;
    for (int l = 0; l < p.L; ++l) {
      int d0, d1;
      float ld;
      ac_src(l, rd, p.D, d0, d1, ld, p.off);
      const float e = __expf((1.f - ld) * s_bl[d0][tid] + ld * s_bl[d1][tid] - mx);
      sum += e;
      num += e * p.disp[l];
    }
    const float pred = num / sum;
    const float g = gpred[((long long)b * p.H + Y) * p.W + X];
    for (int l = 0; l < p.L; ++l) {
      int d0, d1;
      float ld;
      ac_src(l, rd, p.D, d0, d1, ld, p.off);
      const float pr = __expf((1.f - ld) * s_bl[d0][tid] + ld * s_bl[d1][tid] - mx) / sum;
      const float dl = pr * (p.disp[l] - pred) * g;
      s_db[d0][tid] += (1.f - ld) * dl;
      s_db[d1][tid] += ld * dl;
    }
    if (DET)
      for (int d = 0; d < p.D; ++d) vmax = fmaxf(vmax, fabsf(s_db[d][tid]));
  }
  float qs = 0.f;
  double qinv = 0.0;
  if (DET) {                                   // workgroup-wide maximum -> a power-of-two scale with |q| <= 2^50 per contribution
    vmax = dpf_wave_max(vmax);
    if ((tid & 63) == 0) s_max[tid >> 6] = vmax;
    __syncthreads();
    vmax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    if (vmax > 0.f && vmax < 3.0e38f) {
      int e;
      frexpf(vmax, &e);                        // vmax < 2^e
      qs = ldexpf(1.f, 50 - e);
      qinv = ldexp(1.0, e - 50);
    }
  }
  if (active) {
    const float hy = 1.f - ly, hx = 1.f - lx;
    const int ya = y0 - ybase, yb = y1 - ybase, xa = x0 - xbase, xb = x1 - xbase;
    for (int d = 0; d < p.D; ++d) {
      const float v = s_db[d][tid];
      if (DET) {
        const float vs = v * qs;
        atomicAdd(reinterpret_cast<unsigned long long*>(&tile[d][ya][xa]), (unsigned long long)(long long)rintf(hy * hx * vs));
        atomicAdd(reinterpret_cast<unsigned long long*>(&tile[d][ya][xb]), (unsigned long long)(long long)rintf(hy * lx * vs));
        atomicAdd(reinterpret_cast<unsigned long long*>(&tile[d][yb][xa]), (unsigned long long)(long long)rintf(ly * hx * vs));
        atomicAdd(reinterpret_cast<unsigned long long*>(&tile[d][yb][xb]), (unsigned long long)(long long)rintf(ly * lx * vs));
      } else {
        atomicAdd(reinterpret_cast<double*>(&tile[d][ya][xa]), (double)(hy * hx * v));
        atomicAdd(reinterpret_cast<double*>(&tile[d][ya][xb]), (double)(hy * lx * v));
        atomicAdd(reinterpret_cast<double*>(&tile[d][yb][xa]), (double)(ly * hx * v));
        atomicAdd(reinterpret_cast<double*>(&tile[d][yb][xb]), (double)(ly * lx * v));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.D * (TY + 2) * (TX + 2); i += 256) {
    const int xx = i % (TX + 2);
    const int yy = (i / (TX + 2)) % (TY + 2);
    const int d = i / ((TX + 2) * (TY + 2));
    const float v = DET ? (float)((double)tile[d][yy][xx] * qinv) : (float)tile[d][yy][xx];
    const int y = ybase + yy, x = xbase + xx;
    if (v != 0.f && y < p.h && x < p.w) atomicAdd(&dk[(((long long)b * p.D + d) * p.h + y) * p.w + x], v);
  }
}

// backward for the shipped geometry: level interpolation folded at compile time, per-thread values in registers (no LDS arrays)
__global__ __launch_bounds__(256) void head_bwd_8x32_kernel(const float* __restrict__ k, const float* __restrict__ gpred, float* __restrict__ dk, HeadP p) {
  constexpr int D = 8, L = 32;
  constexpr float rd = (float)(D - 1) / (float)(L - 1);
  __shared__ double tile[D][TY + 2][TX + 2];      // fp64 cells: ds_add_f64 (3.1 lanes/clk/CU) instead of ds_add_f32 (0.33)
  const float ry = p.H > 1 ? (float)(p.h - 1) / (float)(p.H - 1) : 0.f;
  const float rx = p.W > 1 ? (float)(p.w - 1) / (float)(p.W - 1) : 0.f;
  const int tilesX = (p.W + TX - 1) / TX, tilesY = (p.H + TY - 1) / TY;
  int bb = blockIdx.x;
  const int tx = bb % tilesX; bb /= tilesX;
  const int ty = bb % tilesY;
  const int b = bb / tilesY;
  const int Y0 = ty * TY, X0 = tx * TX;
  int ybase, xbase, t1;
  float tl;
  ac_src(Y0, ry, p.h, ybase, t1, tl);
  ac_src(X0, rx, p.w, xbase, t1, tl);
  double* flat = &tile[0][0][0];
  for (int i = threadIdx.x; i < D * (TY + 2) * (TX + 2); i += 256) flat[i] = 0.0;
  __syncthreads();
  const int tid = threadIdx.x;
  const int Y = Y0 + (tid >> 5), X = X0 + (tid & 31);
  if (Y < p.H && X < p.W) {
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(Y, ry, p.h, y0, y1, ly);
    ac_src(X, rx, p.w, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const long long lplane = (long long)p.h * p.w;
    const float* kb = k + (long long)b * D * lplane;
    float bl[D], db[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float* q = kb + (long long)d * lplane;
      bl[d] = hy * (hx * q[y0 * p.w + x0] + lx * q[y0 * p.w + x1]) + ly * (hx * q[y1 * p.w + x0] + lx * q[y1 * p.w + x1]);
      db[d] = 0.f;
    }
    float pr[L];
    float mx = -3.4e38f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float src = rd * (float)l;
      int d0 = (int)src;
      if (d0 > D - 1) d0 = D - 1;
      const int d1 = d0 + (d0 < D - 1 ? 1 : 0);
      const float ld = src - (float)d0;
      pr[l] = (1.f - ld) * bl[d0] + ld * bl[d1];
      mx = fmaxf(mx, pr[l]);
    }
    float sum = 0.f, num = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      pr[l] = __expf(pr[l] - mx);
      sum += pr[l];
      num += pr[l] * p.disp[l];
    }
    const float pred = num / sum;
    const float g = gpred[((long long)b * p.H + Y) * p.W + X] / sum;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float src = rd * (float)l;
      int d0 = (int)src;
      if (d0 > D - 1) d0 = D - 1;
      const int d1 = d0 + (d0 < D - 1 ? 1 : 0);
      const float ld = src - (float)d0;
      const float dl = pr[l] * (p.disp[l] - pred) * g;      // softmax-expectation gradient w.r.t. logit l
      db[d0] += (1.f - ld) * dl;
      db[d1] += ld * dl;
    }
    const int ya = y0 - ybase, yb = y1 - ybase, xa = x0 - xbase, xb = x1 - xbase;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = db[d];
      atomicAdd(&tile[d][ya][xa], (double)(hy * hx * v));
      atomicAdd(&tile[d][ya][xb], (double)(hy * lx * v));
      atomicAdd(&tile[d][yb][xa], (double)(ly * hx * v));
      atomicAdd(&tile[d][yb][xb], (double)(ly * lx * v));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D * (TY + 2) * (TX + 2); i += 256) {
    const int xx = i % (TX + 2);
    const int yy = (i / (TX + 2)) % (TY + 2);
    const int d = i / ((TX + 2) * (TY + 2));
    const float v = (float)tile[d][yy][xx];
    const int y = ybase + yy, x = xbase + xx;
    if (v != 0.f && y < p.h && x < p.w) atomicAdd(&dk[(((long long)b * D + d) * p.h + y) * p.w + x], v);
  }
}

}  // namespace

extern "C" {

// logits [B,D,h,w] -> pred [B,H,W], prob [B,L,H,W] (NULL to skip).  disp: L host floats (hypothesis values).
static int softargmin_fwd(const float* logits, float* pred, float* prob, const float* disp_host, int B, int D, int h, int w, int L, int H, int W,
                          float off, hipStream_t st, long long prob_batch_stride = 0) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!logits || !pred || !disp_host || B <= 0 || D <= 0 || D > MAXD || L <= 0 || L > MAXL) return DPF_ERR_INVALID_ARG;
  HeadP p;
  p.B = B; p.D = D; p.h = h; p.w = w; p.L = L; p.H = H; p.W = W; p.off = off; p.pbs = prob_batch_stride;
  for (int i = 0; i < MAXL; ++i) p.disp[i] = i < L ? disp_host[i] : 0.f;
  if (D == 8 && L == 32 && off == 0.f)
    hipLaunchKernelGGL(head_fwd_8x32_kernel, dim3(dpf_ew_grid((long long)B * H * W)), dim3(256), 0, st, logits, pred, prob, p);
  else
    hipLaunchKernelGGL(head_fwd_kernel, dim3(dpf_ew_grid((long long)B * H * W)), dim3(256), 0, st, logits, pred, prob, p);
  return dpf_check_launch();
}

// d logits [B,D,h,w] (zeroed here) from d pred [B,H,W]
static int softargmin_bwd(const float* logits, const float* gpred, float* dlogits, const float* disp_host, int B, int D, int h, int w, int L,
                          int H, int W, float off, hipStream_t st) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!logits || !gpred || !dlogits || !disp_host || B <= 0 || D <= 0 || D > MAXD || L <= 0 || L > MAXL) return DPF_ERR_INVALID_ARG;
  HeadP p;
  p.B = B; p.D = D; p.h = h; p.w = w; p.L = L; p.H = H; p.W = W; p.off = off; p.pbs = 0;
  for (int i = 0; i < MAXL; ++i) p.disp[i] = i < L ? disp_host[i] : 0.f;
  if (hipMemsetAsync(dlogits, 0, sizeof(float) * (size_t)B * D * h * w, st) != hipSuccess) return DPF_ERR_LAUNCH;
  const long long blocks = (long long)B * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
  if (dpf_deterministic()) {
    // phases of tiles with disjoint low-resolution footprints: tiles P apart do not meet when (P * T - T + 1) * ratio >= 2
    const float ry = off > 0.f ? (float)h / (float)H : (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f);
    const float rx = off > 0.f ? (float)w / (float)W : (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f);
    int Py = 1, Px = 1;
    const int tilesY = (H + TY - 1) / TY, tilesX = (W + TX - 1) / TX;
    while (Py < tilesY && (float)(Py * TY - TY + 1) * ry < 2.f) ++Py;
    while (Px < tilesX && (float)(Px * TX - TX + 1) * rx < 2.f) ++Px;
    for (int py = 0; py < Py; ++py)
      for (int px = 0; px < Px; ++px)
        hipLaunchKernelGGL(head_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, logits, gpred, dlogits, p, Py, Px, py, px);
  } else if (D == 8 && L == 32 && off == 0.f)
    hipLaunchKernelGGL(head_bwd_8x32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logits, gpred, dlogits, p);
  else
    hipLaunchKernelGGL(head_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, logits, gpred, dlogits, p, 1, 1, 0, 0);
  return dpf_check_launch();
}

int dpf_softargmin_forward(const float* logits, float* pred, float* prob, const float* disp_host, int B, int D, int h, int w, int L,
                           int H, int W, void* stream) {
  return softargmin_fwd(logits, pred, prob, disp_host, B, D, h, w, L, H, W, 0.f, (hipStream_t)stream);
}
int dpf_softargmin_backward(const float* logits, const float* gpred, float* dlogits, const float* disp_host, int B, int D, int h, int w,
                            int L, int H, int W, void* stream) {
  return softargmin_bwd(logits, gpred, dlogits, disp_host, B, D, h, w, L, H, W, 0.f, (hipStream_t)stream);
}
// the same head with the trilinear upsampling in either convention: align_corners = 0 is F.interpolate(scale_factor=4,
// mode='trilinear', align_corners=False) of src/model/nnet/mainmodel.py:150-153
int dpf_softargmin_forward_ex(const float* logits, float* pred, float* prob, const float* disp_host, int B, int D, int h, int w, int L,
                              int H, int W, int align_corners, void* stream) {
  return softargmin_fwd(logits, pred, prob, disp_host, B, D, h, w, L, H, W, align_corners ? 0.f : 0.5f, (hipStream_t)stream);
}
// forward with the probability volume written at a batch stride (floats): head i of n writes slice [:, i] of a [B, n, L, H, W]
// tensor directly -- the torch.stack of the per-head volumes (modules.py:352-362 + mainmodel.py:98-99) costs no copy
int dpf_softargmin_forward_strided(const float* logits, float* pred, float* prob, long long prob_batch_stride, const float* disp_host, int B,
                                   int D, int h, int w, int L, int H, int W, int align_corners, void* stream) {
  if (prob && prob_batch_stride < (long long)L * H * W) return DPF_ERR_INVALID_ARG;
  return softargmin_fwd(logits, pred, prob, disp_host, B, D, h, w, L, H, W, align_corners ? 0.f : 0.5f, (hipStream_t)stream,
                        prob_batch_stride);
}
int dpf_softargmin_backward_ex(const float* logits, const float* gpred, float* dlogits, const float* disp_host, int B, int D, int h, int w,
                               int L, int H, int W, int align_corners, void* stream) {
  return softargmin_bwd(logits, gpred, dlogits, disp_host, B, D, h, w, L, H, W, align_corners ? 0.f : 0.5f, (hipStream_t)stream);
}

}  // extern "C"
