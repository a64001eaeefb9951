// Training-step tail: masked smooth-L1 (3 heads) + masked "cosine" normal loss, and the fused Adam update over the
// flat parameter arena.
//   loss  : reference src/loss/loss_selector.py:29-42, src/loss/depth/smoothL1.py:15-49 ('given' conversion, target
//           'disp'), src/loss/normal/cosine.py:15-53 (the per-channel, non-summed cosine of SURVEY Q11)
//   Adam  : torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-5) as configured by src/model/model_selector.py:31-34
// One reduction pass over the full-resolution maps (HBM-bound, reads 4+3+1+3+1 planes once), no boolean-mask gather.
#include "dpf_common.h"

namespace {

constexpr int MAXHEADS = 4;
struct LossP {
  int B, n, H, W;
  float wts[MAXHEADS];
  float lam_depth, lam_normal;
};

__device__ __forceinline__ void cosine_terms(const float p[3], const float gt[3], float sim[3], float& pn_norm, float pn[3], float gn[3],
                                             float& den, float& n2, float& m2) {
  const float np = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const float ng = sqrtf(gt[0] * gt[0] + gt[1] * gt[1] + gt[2] * gt[2]);
  pn_norm = fmaxf(np, 1e-6f);
  const float gn_norm = fmaxf(ng, 1e-6f);
#pragma unroll
  for (int c = 0; c < 3; ++c) { pn[c] = p[c] / pn_norm; gn[c] = gt[c] / gn_norm; }
  n2 = sqrtf(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
  m2 = sqrtf(gn[0] * gn[0] + gn[1] * gn[1] + gn[2] * gn[2]);
  den = fmaxf(n2 * m2, 1e-6f);
#pragma unroll
  for (int c = 0; c < 3; ++c) sim[c] = fminf(fmaxf((pn[c] * gn[c]) / den, -1.f), 1.f);
}

// acc[0..n-1] = sum smoothl1 per head, acc[n] = sum (1 - sim), acc[n+1] = count
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ pd, const float* __restrict__ pnorm,
                                                          const float* __restrict__ disp, const float* __restrict__ normal,
                                                          const float* __restrict__ mask, float* __restrict__ acc, LossP p) {
  __shared__ float sm[4];
  const long long hw = (long long)p.H * p.W;
  const long long total = (long long)p.B * hw;
  float s[MAXHEADS] = {0.f, 0.f, 0.f, 0.f};
  float sc = 0.f, cnt = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (!(mask[i] > 0.f)) continue;
    const long long pix = i % hw;
    const long long b = i / hw;
    const float gt = p.n > 0 ? disp[i] : 0.f;
#pragma unroll
    for (int k = 0; k < MAXHEADS; ++k)
      if (k < p.n) {
        const float d = pd[(b * p.n + k) * hw + pix] - gt;
        const float ad = fabsf(d);
        s[k] += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
      }
    if (pnorm) {
      float pv[3], gv[3], sim[3], pn[3], gn[3], nn, den, n2, m2;
#pragma unroll
      for (int c = 0; c < 3; ++c) { pv[c] = pnorm[(b * 3 + c) * hw + pix]; gv[c] = normal[(b * 3 + c) * hw + pix]; }
      cosine_terms(pv, gv, sim, nn, pn, gn, den, n2, m2);
      sc += (1.f - sim[0]) + (1.f - sim[1]) + (1.f - sim[2]);
    }
    cnt += 1.f;
  }
#pragma unroll
  for (int k = 0; k < MAXHEADS; ++k) {
    const float v = dpf_block_sum_256(s[k], sm);
    if (threadIdx.x == 0 && k < p.n) atomicAdd(&acc[k], v);
  }
  sc = dpf_block_sum_256(sc, sm);
  cnt = dpf_block_sum_256(cnt, sm);
  if (threadIdx.x == 0) { atomicAdd(&acc[p.n], sc); atomicAdd(&acc[p.n + 1], cnt); }
}

// out[0] = smoothL1_loss, out[1] = cosine_loss, out[2] = final_loss
__global__ void loss_finalize_kernel(const float* __restrict__ acc, float* __restrict__ out, LossP p, int has_normal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float M = acc[p.n + 1];
  float sl1 = 0.f;
  for (int k = 0; k < p.n; ++k) sl1 += p.wts[k] * (acc[k] / M);
  const float cs = has_normal ? acc[p.n] / (3.f * M) : 0.f;
  out[0] = sl1;
  out[1] = cs;
  out[2] = p.lam_depth * sl1 + p.lam_normal * cs;
}

__global__ void loss_backward_kernel(const float* __restrict__ pd, const float* __restrict__ pnorm, const float* __restrict__ disp,
                                     const float* __restrict__ normal, const float* __restrict__ mask, const float* __restrict__ acc,
                                     const float* __restrict__ gout /*[3] grads of (sl1, cos, final)*/, float* __restrict__ dpd,
                                     float* __restrict__ dpn, LossP p) {
  const long long hw = (long long)p.H * p.W;
  const long long total = (long long)p.B * hw;
  const float M = acc[p.n + 1];
  const float g_sl1 = gout[0] + gout[2] * p.lam_depth;
  const float g_cos = gout[1] + gout[2] * p.lam_normal;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i % hw;
    const long long b = i / hw;
    const bool on = mask[i] > 0.f;
    const float gt = p.n > 0 ? disp[i] : 0.f;
#pragma unroll
    for (int k = 0; k < MAXHEADS; ++k)
      if (k < p.n) {
        float g = 0.f;
        if (on) {
          const float d = pd[(b * p.n + k) * hw + pix] - gt;
          g = g_sl1 * p.wts[k] / M * fminf(fmaxf(d, -1.f), 1.f);
        }
        dpd[(b * p.n + k) * hw + pix] = g;
      }
    if (dpn) {
      float out3[3] = {0.f, 0.f, 0.f};
      if (on) {
        float pv[3], gv[3], sim[3], pn[3], gn[3], nn, den, n2, m2;
#pragma unroll
        for (int c = 0; c < 3; ++c) { pv[c] = pnorm[(b * 3 + c) * hw + pix]; gv[c] = normal[(b * 3 + c) * hw + pix]; }
        cosine_terms(pv, gv, sim, nn, pn, gn, den, n2, m2);
        const float gs = -g_cos / (3.f * M);   // dL/dsim_c
        // q_c = pn_c*gn_c/den, den = max(n2*m2, eps)
        float a[3] = {0.f, 0.f, 0.f};
        const bool den_live = (n2 * m2) > 1e-6f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float q = (pn[c] * gn[c]) / den;
          if (q <= -1.f || q >= 1.f) continue;   // clamp kills the gradient
          a[c] += gs * gn[c] / den;
          if (den_live && n2 > 0.f) {
            const float coef = -gs * (pn[c] * gn[c]) / (den * den) * m2 / n2;
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] += coef * pn[k];
          }
        }
        // pn = p / max(|p|, eps)
        const float np = sqrtf(pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2]);
        if (np > 1e-6f) {
          const float dot = a[0] * pn[0] + a[1] * pn[1] + a[2] * pn[2];
#pragma unroll
          for (int c = 0; c < 3; ++c) out3[c] = (a[c] - pn[c] * dot) / np;
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) out3[c] = a[c] / 1e-6f;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) dpn[(b * 3 + c) * hw + pix] = out3[c];
    }
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float gscale, float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float omb1, float omb2, float eps) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

// the same step with its two step-dependent scalars read from device memory: hyper = { lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) }.  A captured
// HIP graph of the train step replays this launch with fixed arguments; the host refreshes the two floats before each replay.
__global__ void adam_hyper_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                                  float gscale, const float* __restrict__ hyper, float b1, float b2, float omb1, float omb2, float eps) {
  const float lr_over_bc1 = hyper[0], inv_sqrt_bc2 = hyper[1];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_over_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

void fill(LossP& p, int B, int n, int H, int W, const float* w, float l0, float l1) {
  p.B = B; p.n = n; p.H = H; p.W = W;
  for (int i = 0; i < MAXHEADS; ++i) p.wts[i] = i < n ? w[i] : 0.f;
  p.lam_depth = l0; p.lam_normal = l1;
}

}  // namespace

extern "C" {

// pred_depth [B,n,H,W], pred_normal [B,3,H,W] or NULL, disp/mask [B,H,W], normal [B,3,H,W].
// acc_ws: n+2 floats (kept for the backward), out: 3 device floats {smoothL1_loss, cosine_loss, final_loss}.
int dpf_loss_forward(const float* pred_depth, const float* pred_normal, const float* disp, const float* normal, const float* mask,
                     float* acc_ws, float* out, int B, int n, int H, int W, const float* head_weights_host, float lambda_depth,
                     float lambda_normal, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if ((n > 0 && !pred_depth) || (n > 0 && !disp) || !mask || !acc_ws || !out || !head_weights_host || n < 0 || n > MAXHEADS || (pred_normal && !normal)) return DPF_ERR_INVALID_ARG;
  LossP p;
  fill(p, B, n, H, W, head_weights_host, lambda_depth, lambda_normal);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(acc_ws, 0, sizeof(float) * (n + 2), st) != hipSuccess) return DPF_ERR_LAUNCH;
  // deterministic mode: ONE workgroup, so the float atomics that merge the workgroups' partial sums have no partner (dpf_common.h)
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(dpf_deterministic() ? 1 : dpf_ew_grid((long long)B * H * W)), dim3(256), 0, st, pred_depth, pred_normal, disp,
                     normal, mask, acc_ws, p);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, acc_ws, out, p, pred_normal ? 1 : 0);
  return dpf_check_launch();
}

// gout: 3 device floats (upstream gradients of the three outputs); d_pred_depth [B,n,H,W], d_pred_normal [B,3,H,W] or NULL
int dpf_loss_backward(const float* pred_depth, const float* pred_normal, const float* disp, const float* normal, const float* mask,
                      const float* acc_ws, const float* gout, float* d_pred_depth, float* d_pred_normal, int B, int n, int H, int W,
                      const float* head_weights_host, float lambda_depth, float lambda_normal, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if ((n > 0 && (!pred_depth || !disp || !d_pred_depth)) || !mask || !acc_ws || !gout || n < 0 || n > MAXHEADS) return DPF_ERR_INVALID_ARG;
  LossP p;
  fill(p, B, n, H, W, head_weights_host, lambda_depth, lambda_normal);
  hipLaunchKernelGGL(loss_backward_kernel, dim3(dpf_ew_grid((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, pred_depth, pred_normal,
                     disp, normal, mask, acc_ws, gout, d_pred_depth, pred_normal ? d_pred_normal : nullptr, p);
  return dpf_check_launch();
}

// One fused Adam step over a flat arena of n floats; grad is pre-scaled by gscale (1/world_size after an all-reduce SUM).
int dpf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, int step, double lr, double beta1,
                  double beta2, double eps, float gscale, void* stream) {
  dpf_clear_error();   // drop any stale error left by other runtime users (e.g. PyTorch) in this thread
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return DPF_ERR_INVALID_ARG;
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(dpf_ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, gscale,
                     (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
  return dpf_check_launch();
}

// dpf_adam_step with { (float)(lr / (1 - beta1^step)), (float)(1 / sqrt(1 - beta2^step)) } supplied in device memory (hyper[2])
int dpf_adam_step_hyper(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, const float* hyper, double beta1,
                        double beta2, double eps, float gscale, void* stream) {
  dpf_clear_error();
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n <= 0) return DPF_ERR_INVALID_ARG;
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(dpf_ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, gscale, hyper,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
  return dpf_check_launch();
}

}  // extern "C"
