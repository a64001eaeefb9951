"""ORACLE (test infrastructure only) -- CPU restatement of the D3D deformable 3-D convolution.

This file is the *checker* for the HIP deformable-conv kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package (``dualpixelface_amd``) never does.

Parity status: the reference ships no numeric test for this op (``src/module/dcn3d/test.py`` only
prints shapes) and its CUDA extension cannot be built here (CUDA-only ``setup.py``, CPU entry points
are ``AT_ERROR`` stubs) => **parity unpinned by reference tests**.  The restatement is pinned by
known answers instead (tests/test_oracle_dcn.py): zero offsets == ``F.conv3d``; an integer offset ==
a shifted ``F.conv3d``; the explicit backward below == autograd of the forward in fp64.

Algorithm followed (all paths relative to /root/reference/src/module/dcn3d/src/cuda):
  * sampler            deform_im2col_cuda.cuh:26-72   (trilinear, zero outside, per-corner test)
  * im2col + validity  deform_im2col_cuda.cuh:192-265 (sample valid iff -1 < p < size, :248)
  * GEMM / bias        deform_conv_cuda.cu:93-123     (out = bias + W[k, c*27+tap] . col)
  * grad columns       deform_conv_cuda.cu:226-231    (gcol = W^T . grad_out)
  * grad offset        deform_im2col_cuda.cuh:111-190, 336-405
  * grad input         deform_im2col_cuda.cuh:74-109, 267-334
  * grad weight/bias   deform_conv_cuda.cu:254-277
Offset channel layout: channel 3*tap + {0:d, 1:h, 2:w}, tap = i*9 + j*3 + k (cuh:238-240).
``deform_conv3d_forward`` / ``_backward`` restate group == deformable_group == 1 (the only configuration StereoDPNet uses,
normal_module.py:46-51); ``deform_conv3d_forward_grouped`` adds the group logic of deform_conv_cuda.cu:84-121 (one GEMM per conv group over
its slice of the columns) and deform_im2col_cuda.cuh:222-232 (input channel c samples with the offsets of deformable group c // (C / dg)) on
top of the same im2col; it is differentiable torch code, so its gradients come from autograd.
"""
import torch


def _geometry(x, offset, ksize, stride, pad, dil):
    B, C, D, H, W = x.shape
    kd, kh, kw = ksize
    Do = (D + 2 * pad[0] - (dil[0] * (kd - 1) + 1)) // stride[0] + 1
    Ho = (H + 2 * pad[1] - (dil[1] * (kh - 1) + 1)) // stride[1] + 1
    Wo = (W + 2 * pad[2] - (dil[2] * (kw - 1) + 1)) // stride[2] + 1
    T = kd * kh * kw
    assert offset.shape == (B, 3 * T, Do, Ho, Wo), (offset.shape, (B, 3 * T, Do, Ho, Wo))
    dt = x.dtype
    ar = lambda n: torch.arange(n, dtype=dt, device=x.device)
    # integer base coordinate of every (tap, output voxel): cuh:224-226,245-247
    ti = torch.arange(kd).view(kd, 1, 1).expand(kd, kh, kw).reshape(T).to(dt)
    tj = torch.arange(kh).view(1, kh, 1).expand(kd, kh, kw).reshape(T).to(dt)
    tk = torch.arange(kw).view(1, 1, kw).expand(kd, kh, kw).reshape(T).to(dt)
    bd = (ar(Do) * stride[0] - pad[0]).view(1, 1, Do, 1, 1) + (ti * dil[0]).view(1, T, 1, 1, 1)
    bh = (ar(Ho) * stride[1] - pad[1]).view(1, 1, 1, Ho, 1) + (tj * dil[1]).view(1, T, 1, 1, 1)
    bw = (ar(Wo) * stride[2] - pad[2]).view(1, 1, 1, 1, Wo) + (tk * dil[2]).view(1, T, 1, 1, 1)
    off = offset.view(B, T, 3, Do, Ho, Wo)
    pd_ = bd + off[:, :, 0]
    ph_ = bh + off[:, :, 1]
    pw_ = bw + off[:, :, 2]
    return (B, C, D, H, W, T, Do, Ho, Wo), pd_, ph_, pw_


def _corners(pd_, ph_, pw_, D, H, W):
    """8 corners: flat index, validity (sample valid AND corner inside), and the 1-D factors."""
    valid = (pd_ > -1) & (ph_ > -1) & (pw_ > -1) & (pd_ < D) & (ph_ < H) & (pw_ < W)  # cuh:248
    d0 = torch.floor(pd_)
    h0 = torch.floor(ph_)
    w0 = torch.floor(pw_)
    ld, lh, lw = pd_ - d0, ph_ - h0, pw_ - w0
    out = []
    for cd in (0, 1):
        for ch in (0, 1):
            for cw in (0, 1):
                dd, hh, ww = d0 + cd, h0 + ch, w0 + cw
                inside = (dd >= 0) & (dd <= D - 1) & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
                idx = (dd.clamp(0, D - 1) * H + hh.clamp(0, H - 1)) * W + ww.clamp(0, W - 1)
                fd = ld if cd else 1 - ld
                fh = lh if ch else 1 - lh
                fw = lw if cw else 1 - lw
                out.append((idx.long(), valid & inside, (cd, ch, cw), (fd, fh, fw)))
    return out


def deform_im2col(x, offset, ksize, stride, pad, dil):
    """col[B, C, T, Do, Ho, Wo] (cuh:192-265)."""
    (B, C, D, H, W, T, Do, Ho, Wo), pd_, ph_, pw_ = _geometry(x, offset, ksize, stride, pad, dil)
    xf = x.reshape(B, C, D * H * W)
    col = x.new_zeros(B, C, T * Do * Ho * Wo)
    for idx, ok, _, (fd, fh, fw) in _corners(pd_, ph_, pw_, D, H, W):
        wgt = (fd * fh * fw) * ok.to(x.dtype)
        g = torch.gather(xf, 2, idx.reshape(B, 1, -1).expand(B, C, -1))
        col = col + g * wgt.reshape(B, 1, -1)
    return col.view(B, C, T, Do, Ho, Wo)


def deform_conv3d_forward(x, offset, weight, bias, stride=(1, 1, 1), pad=(1, 1, 1), dil=(1, 1, 1)):
    """DCN.deform_conv_forward (deform_conv_cuda.cu:18-126); returns [B, K, Do, Ho, Wo]."""
    K, C, kd, kh, kw = weight.shape
    col = deform_im2col(x, offset, (kd, kh, kw), stride, pad, dil)
    B, _, T, Do, Ho, Wo = col.shape
    out = torch.einsum('kn,bnp->bkp', weight.reshape(K, C * T), col.reshape(B, C * T, -1))
    out = out + bias.view(1, K, 1)
    return out.view(B, K, Do, Ho, Wo)


def deform_conv3d_forward_grouped(x, offset, weight, bias, stride=(1, 1, 1), pad=(1, 1, 1), dil=(1, 1, 1), group=1, deformable_group=1):
    """DCN.deform_conv_forward with group / deformable_group > 1; weight [K, C / group, kd, kh, kw], offset [B, deformable_group * 3 T, ...]."""
    K, Cg, kd, kh, kw = weight.shape
    B, C = x.shape[:2]
    T = kd * kh * kw
    assert C == Cg * group and K % group == 0 and C % deformable_group == 0                       # cu:65-66,75-76
    cd = C // deformable_group
    col = torch.cat([deform_im2col(x[:, d * cd:(d + 1) * cd], offset[:, d * 3 * T:(d + 1) * 3 * T], (kd, kh, kw), stride, pad, dil)
                     for d in range(deformable_group)], 1)                                       # cuh:222,232: [B, C, T, Do, Ho, Wo]
    Do, Ho, Wo = col.shape[3:]
    col = col.reshape(B, group, Cg * T, -1)                                                       # cu:111
    wg = weight.reshape(group, K // group, Cg * T)                                                # cu:84
    out = torch.einsum('gkn,bgnp->bgkp', wg, col).reshape(B, K, -1) + bias.view(1, K, 1)         # cu:113-121
    return out.view(B, K, Do, Ho, Wo)


def deform_conv3d_backward(x, offset, weight, bias, grad_out, stride=(1, 1, 1), pad=(1, 1, 1), dil=(1, 1, 1)):
    """DCN.deform_conv_backward (deform_conv_cuda.cu:128-285) -> grad_input, grad_offset, grad_weight, grad_bias."""
    K, C, kd, kh, kw = weight.shape
    (B, C, D, H, W, T, Do, Ho, Wo), pd_, ph_, pw_ = _geometry(x, offset, (kd, kh, kw), stride, pad, dil)
    P = Do * Ho * Wo
    go = grad_out.reshape(B, K, P)
    gcol = torch.einsum('kn,bkp->bnp', weight.reshape(K, C * T), go).reshape(B, C, T * P)  # cu:226-231
    xf = x.reshape(B, C, D * H * W)
    grad_in = torch.zeros_like(xf)
    g_d = x.new_zeros(B, T * P)
    g_h = x.new_zeros(B, T * P)
    g_w = x.new_zeros(B, T * P)
    for idx, ok, (cd, ch, cw), (fd, fh, fw) in _corners(pd_, ph_, pw_, D, H, W):
        okf = ok.to(x.dtype).reshape(B, 1, -1)
        idxe = idx.reshape(B, 1, -1).expand(B, C, -1)
        # grad_input: adjoint of the sampler (cuh:74-109,313-331)
        wgt = (fd * fh * fw).reshape(B, 1, -1) * okf
        grad_in.scatter_add_(2, idxe, gcol * wgt)
        # grad_offset: d(sample)/d(coord) (cuh:131-187); sign = +1 for the "high" corner
        v = torch.gather(xf, 2, idxe) * okf              # [B, C, T*P]
        s = (v * gcol).sum(1)                             # sum over input channels (cuh:369-401)
        sd = 1.0 if cd else -1.0
        sh = 1.0 if ch else -1.0
        sw = 1.0 if cw else -1.0
        g_d = g_d + sd * (fh * fw).reshape(B, -1) * s
        g_h = g_h + sh * (fd * fw).reshape(B, -1) * s
        g_w = g_w + sw * (fd * fh).reshape(B, -1) * s
    grad_offset = torch.stack([g_d.view(B, T, P), g_h.view(B, T, P), g_w.view(B, T, P)], 2)
    grad_offset = grad_offset.reshape(B, 3 * T, Do, Ho, Wo)
    col = deform_im2col(x, offset, (kd, kh, kw), stride, pad, dil).reshape(B, C * T, P)  # cu:254-261
    grad_w = torch.einsum('bkp,bnp->kn', go, col).view_as(weight)                         # cu:276
    grad_b = go.sum((0, 2))                                                               # cu:277
    return grad_in.view_as(x), grad_offset, grad_w, grad_b


class DeformConv3dFn(torch.autograd.Function):
    """autograd wrapper with the reference's explicit backward (deform_conv_func.py:16-59)."""

    @staticmethod
    def forward(ctx, x, offset, weight, bias, stride, pad, dil):
        ctx.cfg = (tuple(stride), tuple(pad), tuple(dil))
        ctx.save_for_backward(x, offset, weight, bias)
        return deform_conv3d_forward(x, offset, weight, bias, *ctx.cfg)

    @staticmethod
    def backward(ctx, go):
        x, offset, weight, bias = ctx.saved_tensors
        gi, goff, gw, gb = deform_conv3d_backward(x, offset, weight, bias, go.contiguous(), *ctx.cfg)
        return gi, goff, gw, gb, None, None, None
