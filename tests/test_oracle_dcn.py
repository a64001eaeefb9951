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


def _adversarial_offsets(B, D, H, W, seed):
    """Offsets that put the sample coordinate of every (tap, voxel) on an edge case in at least one dimension: exactly -1, inside
    (-1, 0), just below 0, on integers (0, 1, size-1), inside (size-1, size), exactly size, far outside -- mixed with plain
    fractional positions."""
    g = torch.Generator().manual_seed(seed)
    T = 27
    off = torch.zeros(B, 3 * T, D, H, W, dtype=torch.float64)
    dims = (D, H, W)
    for t in range(T):
        tap = (t // 9 - 1, (t // 3) % 3 - 1, t % 3 - 1)
        for ax in range(3):
            n = dims[ax]
            base = torch.arange(n, dtype=torch.float64) + tap[ax]                 # stride 1, pad 1
            shape = [1, 1, 1]
            shape[ax] = n
            base = base.view(shape).expand(D, H, W)
            special = torch.tensor([-1.0, -0.5, -1e-9, 0.0, 0.25, 1.0, n - 1.0, n - 0.5, float(n), n + 3.0, -3.0, 0.999999999, 1.5])
            pick = torch.randint(0, len(special), (B, D, H, W), generator=g)
            target = special[pick]
            plain = torch.rand(B, D, H, W, generator=g, dtype=torch.float64) < 0.35
            target = torch.where(plain, base + torch.randn(B, D, H, W, generator=g, dtype=torch.float64) * 0.8, target)
            off[:, 3 * t + ax] = target - base
    return off


def test_literal_restatement_matches_vectorised_oracle_on_edge_samples():
    """oracle/dcn3d_literal.py (the CUDA kernels walked index by index: (int) truncation, 5x5x5 window, 24-branch coordinate
    weight) and oracle/dcn3d.py (vectorised) are two independent readings of the reference; they must agree to rounding on
    adversarial sample positions -- the strongest pin available for an op whose reference binary cannot be built here."""
    from oracle import dcn3d_literal as lit
    g = torch.Generator().manual_seed(3)
    B, C, K, D, H, W = 1, 2, 3, 2, 3, 4
    x = torch.randn(B, C, D, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, C, 3, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(K, generator=g, dtype=torch.float64)
    for seed in (0, 1):
        off = _adversarial_offsets(B, D, H, W, seed)
        out = deform_conv3d_forward(x, off, w, b)
        go = torch.randn(out.shape, generator=g, dtype=torch.float64)
        out_l = lit.forward(x.numpy(), off.numpy(), w.numpy(), b.numpy())
        assert abs(out.numpy() - out_l).max() < 1e-12
        vec = deform_conv3d_backward(x, off, w, b, go)
        ref = lit.backward(x.numpy(), off.numpy(), w.numpy(), b.numpy(), go.numpy())
        for name, a, e in zip(('grad_input', 'grad_offset', 'grad_weight', 'grad_bias'), vec, ref):
            assert abs(a.numpy() - e).max() < 1e-12, name


def test_grouped_forward_known_answers():
    """deform_conv3d_forward_grouped (group / deformable_group > 1: deform_conv_cuda.cu:84-121, deform_im2col_cuda.cuh:222-232).  Known answers:
    zero offsets == F.conv3d(groups=group) whatever deformable_group is; an integer offset given to ONE deformable group shifts exactly that
    group's input channels (zero fill at the border: the validity rule of cuh:248)."""
    import torch.nn.functional as F
    from oracle import dcn3d
    g = torch.Generator().manual_seed(5)
    B, C, K, D, H, W, group = 2, 8, 6, 3, 5, 6, 2
    x = torch.randn(B, C, D, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, C // group, 3, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(K, generator=g, dtype=torch.float64)
    ref = F.conv3d(x, w, b, padding=1, groups=group)
    for dg in (1, 2, 4):
        off = torch.zeros(B, dg * 81, D, H, W, dtype=torch.float64)
        assert (dcn3d.deform_conv3d_forward_grouped(x, off, w, b, group=group, deformable_group=dg) - ref).abs().max().item() < 1e-12
    # deformable group 1 of 2 (input channels 4 .. 7) samples one voxel further along w
    off = torch.zeros(B, 2 * 81, D, H, W, dtype=torch.float64)
    off[:, 81 + 2::3] = 1.0                              # offset channel 3 tap + 2 = the w coordinate, second deformable group
    xs = x.clone()
    xs[:, 4:, :, :, :-1] = x[:, 4:, :, :, 1:]
    xs[:, 4:, :, :, -1] = 0
    got = dcn3d.deform_conv3d_forward_grouped(x, off, w, b, group=group, deformable_group=2)
    # (column 0 differs by construction: its left tap reads x[0] through the offset, but the padding of the shifted tensor in the plain conv)
    assert (got - F.conv3d(xs, w, b, padding=1, groups=group))[..., 1:].abs().max().item() < 1e-12
