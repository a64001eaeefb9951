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


def deform_conv_forward(input, weight, bias, offset, kernel_d, kernel_h, kernel_w, stride_d, stride_h, stride_w, pad_d, pad_h, pad_w,
                        dilation_d, dilation_h, dilation_w, group, deformable_group, im2col_step):
    _check(input, weight, bias, offset)
    if tuple(weight.shape[2:]) != (kernel_d, kernel_h, kernel_w):
        raise RuntimeError('Input shape and kernel shape wont match')   # deform_conv_cuda.cu:72-73
    try:
        return ops.deform_conv_forward_raw(input.float(), weight.float(), bias.float(), offset.float(), (stride_d, stride_h, stride_w),
                                           (pad_d, pad_h, pad_w), (dilation_d, dilation_h, dilation_w), group, deformable_group,
                                           im2col_step)
    except DpfError as e:
        raise RuntimeError(str(e))


def deform_conv_backward(input, weight, bias, offset, grad_output, kernel_d, kernel_h, kernel_w, stride_d, stride_h, stride_w, pad_d,
                         pad_h, pad_w, dilation_d, dilation_h, dilation_w, group, deformable_group, im2col_step):
    _check(input, weight, bias, offset)
    try:
        return list(ops.deform_conv_backward_raw(input.float(), weight.float(), bias.float(), offset.float(),
                                                 grad_output.float().contiguous(), (stride_d, stride_h, stride_w), (pad_d, pad_h, pad_w),
                                                 (dilation_d, dilation_h, dilation_w), group, deformable_group, im2col_step))
    except DpfError as e:
        raise RuntimeError(str(e))
