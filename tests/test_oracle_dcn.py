"""CPU: known-answer tests that pin the deformable-conv oracle (the reference op is CUDA-only and its own test.py
asserts nothing -- SURVEY section 4 / 8c)."""
import torch
import torch.nn.functional as F

from oracle.dcn3d import deform_conv3d_forward, deform_conv3d_backward, DeformConv3dFn


def _data(dtype=torch.float64):
    g = torch.Generator().manual_seed(0)
    B, C, K, D, H, W = 2, 5, 7, 4, 6, 8
    x = torch.randn(B, C, D, H, W, generator=g, dtype=dtype)
    w = torch.randn(K, C, 3, 3, 3, generator=g, dtype=dtype)
    b = torch.randn(K, generator=g, dtype=dtype)
    return x, w, b, (B, C, K, D, H, W)


def test_zero_offset_is_conv3d():
    x, w, b, (B, C, K, D, H, W) = _data()
    off = torch.zeros(B, 81, D, H, W, dtype=x.dtype)
    assert (deform_conv3d_forward(x, off, w, b) - F.conv3d(x, w, b, padding=1)).abs().max() == 0


def test_integer_offset_is_shifted_conv3d():
    x, w, b, (B, C, K, D, H, W) = _data()
    off = torch.zeros(B, 81, D, H, W, dtype=x.dtype)
    off[:, 1::3] = 1.0                                      # +1 on the H component of every tap
    xp = F.pad(x, (1, 1, 1, 2, 1, 1))
    assert (deform_conv3d_forward(x, off, w, b) - F.conv3d(xp[:, :, :, 1:], w, b)).abs().max() == 0


def test_far_offsets_sample_zero():
    x, w, b, (B, C, K, D, H, W) = _data()
    off = torch.full((B, 81, D, H, W), 100.0, dtype=x.dtype)
    out = deform_conv3d_forward(x, off, w, b)
    assert torch.allclose(out, b.view(1, K, 1, 1, 1).expand_as(out))


def test_explicit_backward_equals_autograd_fp64():
    x, w, b, (B, C, K, D, H, W) = _data()
    g = torch.Generator().manual_seed(1)
    off = torch.randn(B, 81, D, H, W, generator=g, dtype=x.dtype) * 1.5
    x.requires_grad_(); w.requires_grad_(); b.requires_grad_(); off.requires_grad_()
    out = deform_conv3d_forward(x, off, w, b)
    go = torch.randn(out.shape, generator=g, dtype=x.dtype)
    auto = torch.autograd.grad(out, (x, off, w, b), go)
    expl = deform_conv3d_backward(x.detach(), off.detach(), w.detach(), b.detach(), go)
    for a, e in zip(auto, expl):
        assert (a - e).abs().max() < 1e-10


def test_gradcheck_small():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 2, 2, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(2, 2, 3, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(2, generator=g, dtype=torch.float64, requires_grad=True)
    off = (torch.rand(1, 81, 2, 3, 3, generator=g, dtype=torch.float64) * 0.8 + 0.1).requires_grad_()   # away from the floor() kinks
    assert torch.autograd.gradcheck(lambda *a: DeformConv3dFn.apply(*a, (1, 1, 1), (1, 1, 1), (1, 1, 1)), (x, off, w, b), eps=1e-6, atol=1e-5)
