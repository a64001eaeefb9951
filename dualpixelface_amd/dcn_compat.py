"""Drop-in for the reference's pybind module ``DCN`` (src/module/dcn3d/src/vision.cpp:4-7).

    import dualpixelface_amd.dcn_compat as DCN        # or: sys.modules['DCN'] = dualpixelface_amd.dcn_compat
    out = DCN.deform_conv_forward(input, weight, bias, offset, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw,
                                  group, deformable_group, im2col_step)
    grad_input, grad_offset, grad_weight, grad_bias = DCN.deform_conv_backward(input, weight, bias, offset, grad_output, ...same ints...)

Same argument order, tensor layouts and error behaviour as deform_conv.h:10-29,49-69 / deform_conv_cuda.cu:18-285
(contiguous CUDA tensors required -> RuntimeError otherwise; results freshly allocated; runs on the current stream), so
the reference's own ``DeformConvFunction`` (functions/deform_conv_func.py:16-59) can call it unchanged.
"""
import torch

from . import ops
from ._lib import DpfError


def _check(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('input must be a CUDA tensor')          # AT_ASSERTM at deform_conv_cuda.cu:44-47
        if not t.is_contiguous():
            raise RuntimeError('input tensor has to be contiguous')    # deform_conv_cuda.cu:41-42


def _pieces(C, K, T, group, deformable_group):
    """group / deformable_group > 1 (deform_conv_cuda.cu:65-66,84-86,111-121; deform_im2col_cuda.cuh:222-232): the op is a set of independent
    single-group problems -- input channels [c0, c1) in which both the conv group and the offset group are constant, with the output channels
    [k0, k1) of that conv group and the offset channels [o0, o1) of that offset group.  A conv group that spans several offset groups is the
    SUM of its pieces.  The HIP kernels implement group = deformable_group = 1; this module slices, calls them per piece and assembles."""
    if C % group or K % group or C % deformable_group:
        raise RuntimeError('channels(%d) and channels_out(%d) must divide group(%d) / deformable_group(%d)' % (C, K, group, deformable_group))
    cg, cd, kg = C // group, C // deformable_group, K // group
    cuts = sorted(set(list(range(0, C + 1, cg)) + list(range(0, C + 1, cd))))
    out = []
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        g, dg = c0 // cg, c0 // cd
        out.append((c0, c1, g * kg, (g + 1) * kg, dg * 3 * T, (dg + 1) * 3 * T, c0 - g * cg, c1 - g * cg, c0 == g * cg))
    return out


def deform_conv_forward(input, weight, bias, offset, kernel_d, kernel_h, kernel_w, stride_d, stride_h, stride_w, pad_d, pad_h, pad_w,
                        dilation_d, dilation_h, dilation_w, group, deformable_group, im2col_step):
    _check(input, weight, bias, offset)
    if tuple(weight.shape[2:]) != (kernel_d, kernel_h, kernel_w):
        raise RuntimeError('Input shape and kernel shape wont match')   # deform_conv_cuda.cu:72-73
    if input.shape[1] != weight.shape[1] * group:
        raise RuntimeError('Input shape and kernel channels wont match')   # deform_conv_cuda.cu:75-76
    geo = ((stride_d, stride_h, stride_w), (pad_d, pad_h, pad_w), (dilation_d, dilation_h, dilation_w))
    x, w, b, off = input.float(), weight.float(), bias.float(), offset.float()
    try:
        if group == 1 and deformable_group == 1:
            return ops.deform_conv_forward_raw(x, w, b, off, *geo, 1, 1, im2col_step)
        T = kernel_d * kernel_h * kernel_w
        out = None
        for c0, c1, k0, k1, o0, o1, w0, w1, first in _pieces(x.shape[1], w.shape[0], T, group, deformable_group):
            y = ops.deform_conv_forward_raw(x[:, c0:c1].contiguous(), w[k0:k1, w0:w1].contiguous(),
                                            b[k0:k1].contiguous() if first else torch.zeros_like(b[k0:k1]), off[:, o0:o1].contiguous(), *geo, 1, 1,
                                            im2col_step)
            if out is None:
                out = torch.zeros((x.shape[0], w.shape[0]) + tuple(y.shape[2:]), dtype=torch.float32, device=x.device)
            out[:, k0:k1] += y
        return out
    except DpfError as e:
        raise RuntimeError(str(e))


def deform_conv_backward(input, weight, bias, offset, grad_output, kernel_d, kernel_h, kernel_w, stride_d, stride_h, stride_w, pad_d,
                         pad_h, pad_w, dilation_d, dilation_h, dilation_w, group, deformable_group, im2col_step):
    _check(input, weight, bias, offset)
    geo = ((stride_d, stride_h, stride_w), (pad_d, pad_h, pad_w), (dilation_d, dilation_h, dilation_w))
    x, w, b, off, go = input.float(), weight.float(), bias.float(), offset.float(), grad_output.float().contiguous()
    try:
        if group == 1 and deformable_group == 1:
            return list(ops.deform_conv_backward_raw(x, w, b, off, go, *geo, 1, 1, im2col_step))
        T = kernel_d * kernel_h * kernel_w
        gi, goff, gw, gb = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w), torch.zeros_like(b)
        for c0, c1, k0, k1, o0, o1, w0, w1, first in _pieces(x.shape[1], w.shape[0], T, group, deformable_group):
            a, o, ww, bb = ops.deform_conv_backward_raw(x[:, c0:c1].contiguous(), w[k0:k1, w0:w1].contiguous(), b[k0:k1].contiguous(),
                                                        off[:, o0:o1].contiguous(), go[:, k0:k1].contiguous(), *geo, 1, 1, im2col_step)
            gi[:, c0:c1] = a
            goff[:, o0:o1] += o                         # (conv groups inside one offset group add up)
            gw[k0:k1, w0:w1] = ww
            if first:
                gb[k0:k1] = bb
        return [gi, goff, gw, gb]
    except DpfError as e:
        raise RuntimeError(str(e))
