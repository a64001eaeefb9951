"""GPU parity tests (run with ``-m gpu`` on an MI355X): every HIP operator, called through the C ABI, against the CPU
oracle (plain PyTorch-CPU restatements of the reference, oracle/) on the same seeded inputs.

Tolerances: fp32 kernels with different summation orders -> rtol 1e-4 of the tensor's scale (stated per test);
index outputs are bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _ops():
    from dualpixelface_amd import ops
    return ops


def close(a, b, tol=1e-4, name=''):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, '%s: max err %.3e vs scale %.3e (rel %.3e)' % (name, err, scale, err / scale)


def close_elem(a, b, name='', rtol=1e-4, atol=1e-5):
    """Element-wise bound (rtol 1e-4 / atol 1e-5) for operators without a long reduction: every element has to satisfy
    |a - b| <= atol * rms(b) + rtol * |b|  (rms(b) = 1 for unit-scale data, so small-magnitude elements are held to an absolute 1e-5 instead
    of disappearing under the tensor's maximum as in close())."""
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    rms = max(b.pow(2).mean().sqrt().item(), 1e-30)
    excess = (a - b).abs() - (atol * rms + rtol * b.abs())
    worst = excess.max().item()
    assert worst <= 0, '%s: %d of %d elements outside rtol %.0e / atol %.0e x rms %.3e (worst excess %.3e)' % (
        name, int((excess > 0).sum()), excess.numel(), rtol, atol, rms, worst)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # N, C, D, H, W, K, k(3), stride(3), pad(3), dil(3)
    (2, 32, 4, 10, 40, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 64, 8, 16, 24, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (2, 32, 8, 16, 24, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    (1, 35, 4, 9, 33, 81, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (2, 32, 8, 9, 20, 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (2, 32, 3, 12, 36, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    (2, 32, 3, 12, 36, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    (2, 3, 1, 32, 48, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 1, 1)),
    (2, 32, 1, 16, 24, 64, (1, 3, 3), (1, 2, 2), (0, 2, 2), (1, 2, 2)),
    (1, 96, 1, 17, 35, 32, (1, 3, 3), (1, 1, 1), (0, 5, 5), (1, 5, 5)),
    (3, 64, 1, 12, 20, 96, (1, 3, 3), (1, 1, 1), (0, 8, 8), (1, 8, 8)),
    (2, 32, 1, 19, 41, 32, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 3, 3)),      # dilation 3, padding 2: polyphase wgrad, ragged phases
    (1, 32, 2, 40, 70, 32, (1, 3, 3), (1, 1, 1), (0, 4, 4), (1, 4, 4)),      # several tiles per phase
    (2, 128, 1, 6, 9, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    (2, 32, 1, 16, 24, 64, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),
    (4, 32, 1, 8, 12, 3, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    (1, 192, 1, 6, 9, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),      # data gradient has 192 channels (> one launch)
    (1, 40, 2, 5, 7, 160, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    # 16-byte-aligned rows (W % 4 == 0): the LDS-DMA double-buffered kernel (conv_igemm2.hip) -- ragged tiles, channel tails, strides,
    # dilations, two launches over the output channels
    (1, 35, 4, 9, 36, 81, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (2, 32, 3, 37, 72, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 64, 5, 21, 44, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 32, 6, 22, 68, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    (1, 32, 6, 22, 72, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),      # class-fused stride-2 data gradient (g rows 16-byte aligned)
    (2, 48, 7, 13, 24, 33, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),      # odd extents: the data gradient's last plane / row / column has one parity only
    (1, 96, 1, 17, 36, 32, (1, 3, 3), (1, 1, 1), (0, 5, 5), (1, 5, 5)),
    (1, 32, 2, 40, 72, 32, (1, 3, 3), (1, 1, 1), (0, 4, 4), (1, 4, 4)),
    (2, 32, 1, 19, 40, 32, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 3, 3)),
    (2, 64, 1, 45, 48, 64, (1, 3, 3), (1, 1, 1), (0, 8, 8), (1, 8, 8)),      # weight gradient on row-dilated tiles: two blocks of 4 * 8 rows, the second ragged
    (1, 40, 2, 5, 8, 160, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (2, 33, 1, 20, 36, 96, (1, 3, 3), (1, 1, 1), (0, 0, 0), (1, 1, 1)),      # no padding: the data gradient's patch starts outside the tensor
    (1, 192, 1, 9, 12, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    # <= 4 output channels: register-window forward / data-gradient / weight-gradient kernels (conv_smallk.hip)
    (2, 16, 5, 7, 24, 2, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (3, 24, 1, 9, 16, 4, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    (2, 8, 3, 6, 12, 1, (3, 3, 3), (1, 1, 1), (0, 0, 0), (1, 1, 1)),        # no padding: the data gradient's window leaves the g tensor on every side
    (1, 8, 4, 6, 16, 1, (3, 3, 3), (1, 1, 1), (1, 2, 2), (1, 1, 1)),        # padding 2 along H / W
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_backward(case):
    ops = _ops()
    N, C, D, H, W, K, ks, st, pd, dl = case
    x = rnd(N, C, D, H, W, seed=1).requires_grad_()
    w = rnd(K, C, *ks, seed=2, scale=0.1).requires_grad_()
    b = rnd(K, seed=3).requires_grad_()
    y_ref = F.conv3d(x, w, b, st, pd, dl)
    go = rnd(*y_ref.shape, seed=4)
    gx_r, gw_r, gb_r = torch.autograd.grad(y_ref, (x, w, b), go)
    xg, wg, bg = [t.detach().to(DEV).requires_grad_() for t in (x, w, b)]
    y = ops.ConvFn.apply(xg, wg, bg, st, pd, dl)
    close(y, y_ref, 1e-4, 'conv fwd')
    gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), go.to(DEV))
    close(gx, gx_r, 1e-4, 'conv dgrad')
    close(gw, gw_r, 2e-4, 'conv wgrad')
    close(gb, gb_r, 1e-4, 'conv bgrad')


def test_conv_data_gradient_of_leading_channels_only():
    """conv3d(gi_channels=n): the data gradient is computed for the first n input channels only (the rest is zero) -- the constant XYZ
    channels of the ANM volume have no gradient consumer (normal_module.py:166)."""
    ops = _ops()
    x = rnd(2, 35, 3, 10, 24, seed=1)
    w = rnd(81, 35, 3, 3, 3, seed=2, scale=0.1)
    go = rnd(2, 81, 3, 10, 24, seed=3)
    xr = x.clone().requires_grad_()
    (gx_r,) = torch.autograd.grad(F.conv3d(xr, w, None, 1, 1, 1), xr, go)
    xg, wg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    y = ops.conv3d(xg, wg, None, 1, 1, 1, gi_channels=32)
    gx, gw = torch.autograd.grad(y, (xg, wg), go.to(DEV))
    close(gx[:, :32], gx_r[:, :32], 1e-4, 'partial dgrad')
    assert float(gx[:, 32:].abs().max()) == 0.0


@pytest.mark.parametrize('shape', [(2, 64, 2, 4, 6, 64), (1, 64, 4, 8, 12, 32), (2, 64, 1, 3, 5, 32), (1, 64, 3, 5, 8, 32), (2, 35, 2, 6, 36, 64),
                                   (1, 32, 5, 9, 40, 96)])
def test_conv_transpose3d(shape):
    ops = _ops()
    N, Ci, D, H, W, Co = shape
    x = rnd(N, Ci, D, H, W, seed=5).requires_grad_()
    w = rnd(Ci, Co, 3, 3, 3, seed=6, scale=0.1).requires_grad_()
    y_ref = F.conv_transpose3d(x, w, None, 2, 1, 1)
    go = rnd(*y_ref.shape, seed=7)
    gx_r, gw_r = torch.autograd.grad(y_ref, (x, w), go)
    xg, wg = x.detach().to(DEV).requires_grad_(), w.detach().to(DEV).requires_grad_()
    y = ops.conv_transpose3d(xg, wg)
    close(y, y_ref, 1e-4, 'convT fwd')
    gx, gw = torch.autograd.grad(y, (xg, wg), go.to(DEV))
    close(gx, gx_r, 1e-4, 'convT dgrad')
    close(gw, gw_r, 2e-4, 'convT wgrad')


def test_depthwise():
    ops = _ops()
    x = rnd(2, 64, 9, 14, seed=8).requires_grad_()
    w = rnd(64, 1, 3, 3, seed=9).requires_grad_()
    y_ref = F.conv2d(x, w, None, 1, 1, 1, 64)
    go = rnd(*y_ref.shape, seed=10)
    gx_r, gw_r = torch.autograd.grad(y_ref, (x, w), go)
    xg, wg = x.detach().to(DEV).requires_grad_(), w.detach().to(DEV).requires_grad_()
    y = ops.depthwise_conv3x3(xg, wg)
    close_elem(y, y_ref, 'dw fwd')
    gx, gw = torch.autograd.grad(y, (xg, wg), go.to(DEV))
    close_elem(gx, gx_r, 'dw dgrad')
    close(gw, gw_r, 1e-4, 'dw wgrad')


@pytest.mark.parametrize('act', ['none', 'relu', 'prelu'])
@pytest.mark.parametrize('shape', [(2, 32, 7, 9), (3, 64, 4, 6, 10)])
def test_batchnorm_train(act, shape):
    ops = _ops()
    x = (rnd(*shape, seed=11) * 2 + 0.7).requires_grad_()
    C = shape[1]
    w = (torch.rand(C, generator=torch.Generator().manual_seed(12)) + 0.5).requires_grad_()
    b = rnd(C, seed=13).requires_grad_()
    res = rnd(*shape, seed=14).requires_grad_()
    res2 = rnd(*shape, seed=15).requires_grad_()
    slope = torch.tensor([0.17], requires_grad=True)
    rm, rv = rnd(C, seed=16) * 0.1, torch.rand(C, generator=torch.Generator().manual_seed(17)) + 0.5
    rm_r, rv_r = rm.clone(), rv.clone()
    z = F.batch_norm(x, rm_r, rv_r, w, b, True, 0.1, 1e-5) + res
    y_ref = {'none': z, 'relu': F.relu(z), 'prelu': F.prelu(z, slope)}[act] + res2
    go = rnd(*shape, seed=18)
    grads_r = torch.autograd.grad(y_ref, (x, w, b, res, res2) + ((slope,) if act == 'prelu' else ()), go)
    xg, wg, bg, rg, r2g, sg = [t.detach().to(DEV).requires_grad_() for t in (x, w, b, res, res2, slope)]
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    code = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'prelu': ops.ACT_PRELU}[act]
    y = ops.norm_act(xg, wg, bg, sg if act == 'prelu' else None, rg, r2g, rmg, rvg, 1, code)
    close(y, y_ref, 1e-5, 'bn fwd')
    close(rmg, rm_r, 1e-5, 'running_mean')
    close(rvg, rv_r, 1e-5, 'running_var')
    grads = torch.autograd.grad(y, (xg, wg, bg, rg, r2g) + ((sg,) if act == 'prelu' else ()), go.to(DEV))
    for g, gr, nm in zip(grads, grads_r, ('dx', 'dw', 'db', 'dres', 'dres2', 'dslope')):
        close(g, gr, 2e-4, 'bn ' + nm)


class _TwoRankExchange(object):
    """Stands in for distributed.StatExchange on a one-GPU box: plays the other rank of a 2-rank SyncBatchNorm with plain
    torch math, so the local kernels + merge / phase-2 kernels can be checked against full-batch batch norm."""
    world_size = 2

    def __init__(self, x1, dz1, mean_g, invstd_g):
        self.x1, self.dz1, self.mean_g, self.invstd_g = x1, dz1, mean_g, invstd_g
        C = x1.shape[1]
        self.dims = [0] + list(range(2, x1.dim()))
        self.count1 = x1.numel() // C

    def all_gather(self, packed):
        m1 = self.x1.mean(self.dims)
        bshape = [1, -1] + [1] * (self.x1.dim() - 2)
        M2 = ((self.x1 - m1.view(bshape)) ** 2).sum(self.dims)
        other = torch.cat([torch.stack([m1, M2], 1).reshape(-1), torch.tensor([float(self.count1)], device=packed.device)])
        return torch.stack([packed, other.to(packed.dtype)])

    def all_reduce_sum_(self, ws):
        bshape = [1, -1] + [1] * (self.x1.dim() - 2)
        xh = (self.x1 - self.mean_g.view(bshape)) * self.invstd_g.view(bshape)
        C = self.x1.shape[1]
        v = ws[:3 * C].view(-1, 3)
        v[:, 0] += self.dz1.sum(self.dims)
        v[:, 1] += (self.dz1 * xh).sum(self.dims)
        ws[3 * C] += float(self.count1)           # the other rank's element count rides in the last slot
        return ws


@pytest.mark.parametrize('shape', [(2, 32, 7, 9), (2, 16, 4, 6, 12)])
def test_sync_batchnorm_matches_full_batch(shape):
    """rank 0 holds the first 2 samples, the emulated rank 1 three more: outputs, running statistics and dx must equal batch norm
    over all 5 samples (torch.nn.SyncBatchNorm semantics); dw/db are this rank's partial sums (DDP averages them afterwards)."""
    ops = _ops()
    C = shape[1]
    full = (5,) + tuple(shape[1:])
    x = (rnd(*full, seed=21) * 1.5 - 0.3).requires_grad_()
    w = (torch.rand(C, generator=torch.Generator().manual_seed(22)) + 0.5).requires_grad_()
    b = rnd(C, seed=23).requires_grad_()
    rm, rv = rnd(C, seed=24) * 0.1, torch.rand(C, generator=torch.Generator().manual_seed(25)) + 0.5
    rm_r, rv_r = rm.clone(), rv.clone()
    z = F.batch_norm(x, rm_r, rv_r, w, b, True, 0.1, 1e-5)
    y_ref = F.relu(z)
    go = rnd(*full, seed=26)
    gx_r, = torch.autograd.grad(y_ref, (x,), go)
    dims = [0] + list(range(2, x.dim()))
    bshape = [1, -1] + [1] * (x.dim() - 2)
    mean_g = x.detach().mean(dims)
    invstd_g = 1.0 / torch.sqrt(x.detach().var(dims, unbiased=False) + 1e-5)
    dz = (go * (z.detach() > 0)).detach()
    xh = (x.detach() - mean_g.view(bshape)) * invstd_g.view(bshape)
    ex = _TwoRankExchange(x.detach()[2:].to(DEV), dz[2:].to(DEV), mean_g.to(DEV), invstd_g.to(DEV))
    xg, wg, bg = [t.detach().to(DEV).requires_grad_() for t in (x[:2], w, b)]
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    y = ops.norm_act(xg, wg, bg, None, None, None, rmg, rvg, 1, ops.ACT_RELU, exchange=ex)
    close(y, y_ref[:2], 1e-5, 'syncbn fwd')
    close(rmg, rm_r, 1e-5, 'syncbn running_mean')
    close(rvg, rv_r, 1e-5, 'syncbn running_var')
    gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), go[:2].to(DEV))
    close(gx, gx_r[:2], 2e-4, 'syncbn dx')
    close(gw, (dz * xh)[:2].sum(dims), 2e-4, 'syncbn dw (local part)')
    close(gb, dz[:2].sum(dims), 2e-4, 'syncbn db (local part)')


def test_batchnorm_eval_and_instance_norm_and_leaky():
    ops = _ops()
    x = rnd(2, 32, 3, 6, 8, seed=20)
    w, b = torch.rand(32) + 0.5, rnd(32, seed=21)
    rm, rv = rnd(32, seed=22) * 0.2, torch.rand(32) + 0.5
    y_ref = F.relu(F.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5))
    y = ops.norm_act(x.to(DEV), w.to(DEV), b.to(DEV), None, None, None, rm.to(DEV), rv.to(DEV), 2, ops.ACT_RELU)
    close(y, y_ref, 1e-5, 'bn eval')
    # instance norm + sigmoid (asm.py:138,154,162)
    xr = x.clone().requires_grad_()
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = torch.sigmoid(F.instance_norm(xr, None, None, wr, br, True, 0.1, 1e-5))
    go = rnd(*x.shape, seed=23)
    gr = torch.autograd.grad(y_ref, (xr, wr, br), go)
    xg, wg, bg = [t.detach().to(DEV).requires_grad_() for t in (x, w, b)]
    y = ops.norm_act(xg, wg, bg, mode=3, act=ops.ACT_SIGMOID)
    close(y, y_ref, 1e-5, 'in fwd')
    gg = torch.autograd.grad(y, (xg, wg, bg), go.to(DEV))
    for a, r, nm in zip(gg, gr, ('dx', 'dw', 'db')):
        close(a, r, 2e-4, 'in ' + nm)
    # stand-alone LeakyReLU(0.1) and a plain add
    xr = x.clone().requires_grad_()
    y_ref = F.leaky_relu(xr, 0.1)
    (gr,) = torch.autograd.grad(y_ref, xr, go)
    xg = x.to(DEV).requires_grad_()
    y = ops.norm_act(xg, act=ops.ACT_LEAKY, slope_const=0.1)
    close_elem(y, y_ref, 'leaky')
    (gg,) = torch.autograd.grad(y, xg, go.to(DEV))
    close_elem(gg, gr, 'leaky bwd')


@pytest.mark.parametrize('scale', [2, 4])
def test_bilinear_and_nearest(scale):
    ops = _ops()
    x = rnd(2, 5, 7, 11, seed=30).requires_grad_()
    y_ref = F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=True)
    go = rnd(*y_ref.shape, seed=31)
    (gr,) = torch.autograd.grad(y_ref, x, go)
    xg = x.detach().to(DEV).requires_grad_()
    y = ops.upsample_bilinear(xg, scale)
    close_elem(y, y_ref, 'bilinear fwd')
    (gg,) = torch.autograd.grad(y, xg, go.to(DEV))
    close_elem(gg, gr, 'bilinear bwd')
    lat = rnd(2, 5, 7 * scale, 11 * scale, seed=32).requires_grad_()
    top = rnd(2, 5, 7, 11, seed=33).requires_grad_()
    y_ref = lat + F.interpolate(top, size=lat.shape[-2:], mode='nearest')
    gl_r, gt_r = torch.autograd.grad(y_ref, (lat, top), go)
    lg, tg = lat.detach().to(DEV).requires_grad_(), top.detach().to(DEV).requires_grad_()
    y = ops.nearest_up_add(lg, tg)
    close_elem(y, y_ref, 'nearest fwd')
    gl, gt = torch.autograd.grad(y, (lg, tg), go.to(DEV))
    close_elem(gl, gl_r, 'nearest dlat')
    close_elem(gt, gt_r, 'nearest dtop')


def _device_tables(h, w, delta):
    from dualpixelface_amd.sampler_tables import build_phase_tables, build_shift_tables, is_fractional
    tables = tuple(t.to(DEV) for t in build_shift_tables(h, w, delta))
    phase = None
    if is_fractional(delta):
        phase = tuple(t.to(DEV) if torch.is_tensor(t) else t for t in build_phase_tables(h, w, delta))
    return tables, phase


@pytest.mark.parametrize('delta', [-1.0, 1.0, 2.0, 0.5, -0.25, 1.75, -2.5])
@pytest.mark.parametrize('shape', [(2, 8, 12, 20), (1, 3, 40, 70), (1, 2, 256, 96)])
def test_shift_triple(delta, shape):
    """nearest / bilinear / Fourier-phase triple incl. fractional shifts (row circulant on MFMA + Hilbert term) and its adjoint."""
    from oracle.stereodpnet import StereoDPNetOracle
    ops = _ops()
    fea = rnd(*shape, seed=40).requires_grad_()
    ref = torch.stack(StereoDPNetOracle.shift_triple(fea, delta), 2)
    go = rnd(*ref.shape, seed=41)
    (gr,) = torch.autograd.grad(ref, fea, go)
    tables, phase = _device_tables(shape[2], shape[3], delta)
    fg = fea.detach().to(DEV).requires_grad_()
    out = ops.shift_triple(fg, tables, phase)
    close_elem(out, ref, 'shift fwd')
    (gg,) = torch.autograd.grad(out, fg, go.to(DEV))
    close_elem(gg, gr, 'shift bwd')
    # the adjoint is a gather (no atomics): bitwise reproducible
    (gg2,) = torch.autograd.grad(ops.shift_triple(fg, tables, phase), fg, go.to(DEV))
    assert torch.equal(gg, gg2)


@pytest.mark.parametrize('delta', [1.0, 0.5])
def test_shift_triple_many_planes(delta):
    """B*C*3 > 65535 planes (the plane index rides in gridDim.y): the launchers slice; every plane still equals the oracle's."""
    from oracle.stereodpnet import StereoDPNetOracle
    ops = _ops()
    shape = (2, 33000, 4, 8)                               # 66 000 (b, c) planes, 198 000 output planes
    fea = rnd(*shape, seed=44)
    tables, phase = _device_tables(shape[2], shape[3], delta)
    fg = fea.to(DEV).requires_grad_()
    out = ops.shift_triple(fg, tables, phase)
    sel = [0, 1, 21844, 21845, 21846, 32999]               # around the slice seams
    ref = torch.stack(StereoDPNetOracle.shift_triple(fea[:, sel].clone().requires_grad_(), delta), 2)
    close_elem(out[:, sel], ref, 'shift fwd, many planes')
    go = rnd(*out.shape, seed=45)
    (gg,) = torch.autograd.grad(out, fg, go.to(DEV))
    fs = fea[:, sel].clone().requires_grad_()
    (gr,) = torch.autograd.grad(torch.stack(StereoDPNetOracle.shift_triple(fs, delta), 2), fs, go[:, sel])
    close_elem(gg[:, sel], gr, 'shift bwd, many planes')
    # second half of the batch lives past plane 65535 of the adjoint's grid too
    assert torch.isfinite(gg).all() and gg[1].abs().sum().item() > 0


def test_shift_triple_fractional_vs_reference_fixture(golden_dir):
    """Fractional shifts against outputs of the reference's own subpixel_shift (fresh module instance per delta)."""
    ops = _ops()
    g = np.load(golden_dir + '/shift_fractional.npz')
    for ci in range(3):
        fea = torch.from_numpy(g['fea%d' % ci]).to(DEV)
        for di, delta in enumerate(g['deltas']):
            for direction, sign in (('forward', 1.0), ('backward', -1.0)):
                tables, phase = _device_tables(fea.shape[2], fea.shape[3], sign * float(delta))
                out = ops.shift_triple(fea, tables, phase)
                for j, nm in enumerate(('nearest', 'bilinear', 'phase')):
                    close(out[:, :, j], torch.from_numpy(g['c%d_d%d_%s_%s' % (ci, di, direction, nm)]), 5e-6, '%s %s %s' % (nm, delta, direction))


def test_cv_select():
    ops = _ops()
    B, C, h, w, L = 2, 8, 6, 10, 8
    ts = [rnd(B, C, 3, h, w, seed=50 + i).requires_grad_() for i in range(4)]
    ts[1] = torch.sigmoid(ts[1].detach()).requires_grad_()
    ts[3] = torch.sigmoid(ts[3].detach()).requires_grad_()

    def sel(x3, s):
        return torch.mean(x3 * F.softmax(s, dim=2), 2)
    ref = torch.cat([sel(ts[0], ts[1]), sel(ts[2], ts[3])], 1).unsqueeze(2).expand(-1, -1, L, -1, -1)
    go = rnd(B, 2 * C, L, h, w, seed=55)
    gr = torch.autograd.grad(ref, ts, go)
    tg = [t.detach().to(DEV).requires_grad_() for t in ts]
    vol = ops.cv_select(L, [(1 << L) - 1], tg)
    close_elem(vol, ref, 'cv_select fwd')
    gg = torch.autograd.grad(vol, tg, go.to(DEV))
    for a, r in zip(gg, gr):
        close_elem(a, r, 'cv_select bwd')


def test_softargmin():
    ops = _ops()
    B, D, h, w = 2, 8, 6, 10
    logits = rnd(B, 1, D, h, w, seed=60, scale=2.0).requires_grad_()
    disp = [-4 + 0.5 * i for i in range(32)]
    up = F.interpolate(logits, scale_factor=4, mode='trilinear', align_corners=True).squeeze(1)
    prob_r = F.softmax(up, 1)
    pred_r = torch.sum(prob_r * torch.tensor(disp).view(1, -1, 1, 1), 1)
    go = rnd(*pred_r.shape, seed=61)
    (gr,) = torch.autograd.grad(pred_r, logits, go)
    lg = logits.detach().to(DEV).requires_grad_()
    pred, prob = ops.softargmin(lg, disp, 4, True)
    close(pred, pred_r, 1e-5, 'softargmin pred')
    close(prob, prob_r, 1e-5, 'softargmin prob')
    (gg,) = torch.autograd.grad(pred, lg, go.to(DEV))
    close(gg, gr, 1e-4, 'softargmin bwd')


@pytest.mark.parametrize('cfg', [(2, 35, 64, 4, 6, 9), (1, 64, 64, 4, 8, 12), (2, 5, 7, 3, 5, 6),
                                 # several tiles per axis and offsets reaching past the staged halo (slow paths), 12- and 16-wide chunks
                                 (1, 20, 40, 5, 9, 70, 3.0), (1, 16, 24, 4, 7, 45, 4.0),
                                 # the lean-sampler kernels' domain (depth <= 4, W % 4 == 0): the model's channel counts, several tiles, wide
                                 # offsets (cooperative slow path), shallow volumes, a partial tile row
                                 (2, 35, 64, 4, 6, 12), (1, 20, 40, 4, 9, 72, 3.0), (1, 16, 24, 3, 7, 44, 4.0), (1, 36, 33, 2, 37, 20, 6.0),
                                 (1, 12, 8, 1, 5, 8, 1.0),
                                 # channel counts whose lean BACKWARD weight repack (27 x ceil(C / 12) x 1024 floats) is larger than the region
                                 # kernels' repack (27 x pad32(C) x pad64(K)): the grad_weight replicas must start behind it
                                 (1, 60, 24, 2, 6, 12), (1, 84, 16, 3, 5, 8, 2.0)])
def test_deform_conv(cfg):
    from oracle import dcn3d
    ops = _ops()
    B, C, K, D, H, W = cfg[:6]
    x = rnd(B, C, D, H, W, seed=70)
    off = rnd(B, 81, D, H, W, seed=71, scale=cfg[6] if len(cfg) > 6 else 1.5)
    wt = rnd(K, C, 3, 3, 3, seed=72, scale=0.1)
    bs = rnd(K, seed=73)
    y_ref = dcn3d.deform_conv3d_forward(x, off, wt, bs)
    go = rnd(*y_ref.shape, seed=74)
    gr = dcn3d.deform_conv3d_backward(x, off, wt, bs, go)
    xg, og, wg, bg = [t.to(DEV).requires_grad_() for t in (x, off, wt, bs)]
    y = ops.deform_conv3d(xg, og, wg, bg)
    close(y, y_ref, 1e-4, 'dcn fwd')
    gg = torch.autograd.grad(y, (xg, og, wg, bg), go.to(DEV))
    for a, r, nm in zip(gg, gr, ('grad_input', 'grad_offset', 'grad_weight', 'grad_bias')):
        close(a, r, 2e-4, 'dcn ' + nm)
    # known answer: zero offsets == plain conv3d
    y0 = ops.deform_conv3d(xg, torch.zeros_like(og), wg, bg)
    close(y0, F.conv3d(x, wt, bs, padding=1), 1e-4, 'dcn zero offset')


@pytest.mark.parametrize('cfg', [(2, 32, 32, 4, 16, 64, (3, 3, 3), 1, 1), (2, 24, 40, 1, 24, 96, (1, 3, 3), 1, 3), (1, 32, 64, 4, 16, 32, (3, 3, 3), 2, 1)])
def test_weight_gradient_f32_matrix_paths_agree(cfg):
    """The weight gradient multiplies fp32 operands either on v_mfma_f32_32x32x2_f32 or as nine exact bf16 partial products on the bf16
    matrix pipe (dpf_set_f32_matrix_path).  Both are fp32 arithmetic: against an fp64 reference neither may be worse than 2 x the other,
    and each is within 1e-5 of the tensor scale."""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, stride, dil = cfg
    pad = tuple(((k - 1) * dil) // 2 if k > 1 else 0 for k in ks)
    dl = tuple(dil if k > 1 else 1 for k in ks)
    st = (stride if ks[0] > 1 else 1, stride, stride)
    x = rnd(N, C, D, H, W, seed=180)
    od = [(i + 2 * p - (d * (k - 1) + 1)) // s + 1 for i, k, s, p, d in zip((D, H, W), ks, st, pad, dl)]
    g = rnd(N, K, *od, seed=181)
    ref = torch.nn.grad.conv3d_weight(x.double(), (K, C) + ks, g.double(), st, pad, dl)
    prev = lib().cdll.dpf_get_f32_matrix_path()          # (a getter: the value is not an error code)
    errs = []
    try:
        for path in (1, 0, 2):
            lib().call('dpf_set_f32_matrix_path', path)
            got = ops._conv_wgrad_raw(g.to(DEV), x.to(DEV), (K, C) + ks, st, pad, dl).double().cpu()
            errs.append(((got - ref).abs().max() / ref.abs().max()).item())
        # path 2 (two f16 components, tiles converted in place and scaled by the running maxima): power-of-two scaling of either operand
        # is exact, far outside the f16 range, and a batch whose samples are 2^30 apart is summed as accurately as on the fp32 instruction
        # (launches with fewer than five column tiles -- the stride-2 configuration -- stay on the six bf16 products: conv_wgrad2.hip)
        base = ops._conv_wgrad_raw(g.to(DEV), x.to(DEV), (K, C) + ks, st, pad, dl)
        for kg, kx in ((-90, 0), (40, -80), (0, 70), (-45, -45)) if stride == 1 else ():
            got = ops._conv_wgrad_raw((g * 2.0 ** kg).to(DEV), (x * 2.0 ** kx).to(DEV), (K, C) + ks, st, pad, dl)
            assert torch.equal(got, base * 2.0 ** (kg + kx)), (kg, kx)
        g2 = g.clone()
        g2[0] *= 2.0 ** -30                    # the small sample comes first: the running exponent grows when the second one arrives
        ref2 = torch.nn.grad.conv3d_weight(x.double(), (K, C) + ks, g2.double(), st, pad, dl)
        e = []
        for path in (2, 0):
            lib().call('dpf_set_f32_matrix_path', path)
            got = ops._conv_wgrad_raw(g2.to(DEV), x.to(DEV), (K, C) + ks, st, pad, dl).double().cpu()
            e.append(((got - ref2).abs().max() / ref2.abs().max()).item())
        assert e[0] <= 2 * e[1] + 1e-7, e
    finally:
        lib().call('dpf_set_f32_matrix_path', prev)
    assert max(errs) <= 1e-5, errs
    assert errs[0] <= 2 * errs[1] + 1e-7 and errs[1] <= 2 * errs[0] + 1e-7, errs
    assert errs[2] <= 2 * errs[1] + 1e-7, errs


@pytest.mark.parametrize('cfg', [
    # N, C, K, D, H, W, kernel, dilation, transposed (data gradient)
    (1, 64, 32, 6, 20, 40, (3, 3, 3), 1, False),     # 4-channel chunks, tap quadruples, four output planes per tile
    (2, 32, 64, 3, 18, 36, (3, 3, 3), 1, True),      # two row tiles, one weight buffer
    (1, 35, 81, 4, 12, 36, (3, 3, 3), 1, False),     # channel tail (35 = 8 x 4 + 3), output channels in slices of 64 + 17
    (2, 96, 32, 1, 24, 72, (1, 3, 3), 1, False),     # 8-channel chunks, tap pairs (9 taps -> 5 groups)
    (1, 40, 96, 1, 21, 40, (1, 3, 3), 2, True),      # dilation 2, ragged rows, 64 + 32 output channels
])
def test_conv_f32_matrix_paths_agree(cfg):
    """Stride-1 convolutions multiply fp32 operands either on v_mfma_f32_32x32x2_f32 (igemm2_kernel) or as exact bf16 partial products on
    the bf16 matrix pipe (igemm3_x9_kernel: every operand split exactly into three bf16 terms, eight of the nine partial products -- the
    dropped lo x lo term is below 2^-32 of a product).  Both are fp32 arithmetic: against an fp64 reference neither may be worse than
    2 x the other, each is within 1e-5 of the tensor scale -- also on all-positive data, where a truncation bias would add up."""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, dil, transposed = cfg
    pad = tuple(((k - 1) * dil) // 2 if k > 1 else 0 for k in ks)
    dl = tuple(dil if k > 1 else 1 for k in ks)
    prev = lib().cdll.dpf_get_f32_matrix_path()
    try:
        for positive in (False, True):
            x = rnd(N, C, D, H, W, seed=190)
            w = rnd(K, C, *ks, seed=191, scale=0.1)
            if positive:
                x, w = x.abs(), w.abs()
            if transposed:
                g = rnd(N, K, D, H, W, seed=192)
                g = g.abs() if positive else g
                ref = torch.nn.grad.conv3d_input((N, C, D, H, W), w.double(), g.double(), 1, pad, dl)
            else:
                ref = F.conv3d(x.double(), w.double(), None, 1, pad, dl)
            errs = []
            for path in (1, 0, 2):
                lib().call('dpf_set_f32_matrix_path', path)
                if transposed:
                    xg = x.to(DEV).requires_grad_()
                    y = ops.ConvFn.apply(xg, w.to(DEV), None, (1, 1, 1), pad, dl)
                    (got,) = torch.autograd.grad(y, xg, g.to(DEV))
                else:
                    got = ops.ConvFn.apply(x.to(DEV), w.to(DEV), None, (1, 1, 1), pad, dl)
                errs.append(((got.double().cpu() - ref).abs().max() / ref.abs().max()).item())
            assert max(errs) <= 1e-5, (positive, errs)
            assert errs[0] <= 2 * errs[1] + 1e-7 and errs[1] <= 2 * errs[0] + 1e-7, (positive, errs)
            assert errs[2] <= 2 * errs[1] + 1e-7, (positive, errs)           # the f16 components: no worse than 2 x the fp32 instruction
    finally:
        lib().call('dpf_set_f32_matrix_path', prev)


@pytest.mark.parametrize('cfg', [(2, 32, 32, 4, 24, 64, (3, 3, 3), False), (2, 48, 64, 1, 20, 72, (1, 3, 3), True)])
def test_conv_f16_component_path_block_scaling(cfg):
    """dpf_set_f32_matrix_path(2): two f16 components per operand, scaled per block (a channel chunk of a tile's patch; the weight tensor) by a
    power of two.  (a) The scaling is exact: multiplying the input by 2^k and the weights by 2^j multiplies the output by 2^(k+j) BIT FOR
    BIT, from 2^-100 to 2^+100 -- no overflow, no flush.  (b) Two samples that differ by 2^40 in magnitude: each is as accurate (relative to its
    own scale) as with the fp32 instruction -- the scale follows the tile, not the tensor.  (c) zeros, and a single
    non-zero value in an otherwise zero tensor."""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, transposed = cfg
    pad = tuple((k - 1) // 2 for k in ks)
    one = (1, 1, 1)
    prev = lib().cdll.dpf_get_f32_matrix_path()

    def run(x, w):
        if transposed:                      # the data gradient of a conv with weights w [K, C] for the output gradient x [N, K]
            xin = torch.zeros(N, C, D, H, W, device=DEV).requires_grad_()
            y = ops.ConvFn.apply(xin, w.to(DEV), None, one, pad, one)
            return torch.autograd.grad(y, xin, x.to(DEV))[0]
        return ops.ConvFn.apply(x.to(DEV), w.to(DEV), None, one, pad, one)

    try:
        lib().call('dpf_set_f32_matrix_path', 2)
        x = rnd(N, K if transposed else C, D, H, W, seed=300)
        w = rnd(K, C, *ks, seed=301, scale=0.1)
        base = run(x, w)
        for k, j in ((-100, 0), (100, -60), (0, 90), (-50, -50), (30, 30)):
            got = run(x * 2.0 ** k, w * 2.0 ** j)
            assert torch.equal(got, base * 2.0 ** (k + j)), (k, j)
        # (b) two samples 2^40 apart in magnitude (a tile never spans two samples)
        x2 = x.clone()
        x2[0] *= 2.0 ** -40
        lib().call('dpf_set_f32_matrix_path', 0)
        f32 = run(x2, w).double().cpu()
        lib().call('dpf_set_f32_matrix_path', 2)
        got = run(x2, w).double().cpu()
        if transposed:
            ref = torch.nn.grad.conv3d_input((N, C, D, H, W), w.double(), x2.double(), 1, pad, 1)
        else:
            ref = F.conv3d(x2.double(), w.double(), None, 1, pad, 1)
        for n in range(N):
            scale = ref[n].abs().max()
            e2, e0 = (got - ref)[n].abs().max() / scale, (f32 - ref)[n].abs().max() / scale
            assert e2 <= 2 * e0 + 1e-7, (n, e2.item(), e0.item())
        # (c)
        z = torch.zeros_like(x)
        assert run(z, w).abs().max().item() == 0.0
        z[0, 3, 0, 5, 7] = 3.0e-30
        lib().call('dpf_set_f32_matrix_path', 0)
        a = run(z, w)
        lib().call('dpf_set_f32_matrix_path', 2)
        b = run(z, w)
        assert (a - b).abs().max().item() <= 3e-7 * a.abs().max().item() and a.abs().max().item() > 0
    finally:
        lib().call('dpf_set_f32_matrix_path', prev)


RANGE_EXPS = (0, 13, 20, 27, 34)            # binary orders below the block maximum


def _spread(t, layout, seed):
    """t scaled by 2^-e, e from RANGE_EXPS, laid out so that ONE tile / channel chunk of the kernels holds all five magnitudes:
    'wbands' / 'hbands': bands of 6 columns / 3 rows (an output inside a band sees only that band's magnitude), 'chan': per channel
    (c % 5: every 8-channel chunk holds all five), 'elem': per element at random."""
    e = torch.tensor(RANGE_EXPS, dtype=torch.float64)
    N, C, D, H, W = t.shape
    if layout == 'wbands':
        idx = (torch.arange(W) // 6) % 5
        sc = e[idx].view(1, 1, 1, 1, W)
    elif layout == 'hbands':
        idx = (torch.arange(H) // 3) % 5
        sc = e[idx].view(1, 1, 1, H, 1)
    elif layout == 'chan':
        idx = torch.arange(C) % 5
        sc = e[idx].view(1, C, 1, 1, 1)
    else:
        g = torch.Generator().manual_seed(seed)
        sc = e[torch.randint(0, 5, t.shape, generator=g)]
    return (t.double() * torch.pow(2.0, -sc)).float()


@pytest.mark.parametrize('cfg', [
    # N, C, K, D, H, W, kernel, dilation, transposed (data gradient)
    (1, 32, 32, 4, 24, 64, (3, 3, 3), 1, False),     # 4-channel chunks (the hourglass layers)
    (1, 32, 64, 3, 18, 64, (3, 3, 3), 1, True),
    (2, 48, 32, 1, 24, 64, (1, 3, 3), 1, False),     # 8-channel chunks (2-D layers)
    (1, 40, 64, 1, 24, 64, (1, 3, 3), 2, True),      # dilated rows
])
@pytest.mark.parametrize('layout', ['wbands', 'hbands', 'chan', 'elem', 'wrows'])
def test_conv_f16_component_path_in_block_dynamic_range(cfg, layout):
    """dpf_set_f32_matrix_path(2) against the fp32 matrix instruction (path 0) when ONE tile / channel chunk holds values at 1, 2^-13, 2^-20,
    2^-27 and 2^-34 of its maximum -- signed and all-positive data, forward (the spread in x) and data gradient (the spread in g).  Every
    output element is compared with fp64 relative to ITS OWN magnitude: the error is divided by sum |w| |x| over the element's own window
    (for all-positive data that is the element itself), so an output that only sees 2^-34-sized inputs has to be right to fp32 precision of
    2^-34-sized numbers.  'wrows' puts the spread into the weights instead: output row r (an output channel; for the data gradient an
    input channel) at 2^-RANGE_EXPS[r % 5] -- the pack kernel scales every row by its own exponent.  Bar: the worst element of path 2
    within 2 x the worst element of path 0.  Negative control: with the range guard
    switched off (block scaling without residual passes -- round 5's kernel) the banded layouts miss the bar by orders of magnitude."""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, dil, transposed = cfg
    pad = tuple(((k - 1) * dil) // 2 if k > 1 else 0 for k in ks)
    dl = tuple(dil if k > 1 else 1 for k in ks)
    one = (1, 1, 1)
    prev = lib().cdll.dpf_get_f32_matrix_path()

    def run(a, w):
        if transposed:
            xin = torch.zeros(N, C, D, H, W, device=DEV).requires_grad_()
            y = ops.ConvFn.apply(xin, w.to(DEV), None, one, pad, dl)
            return torch.autograd.grad(y, xin, a.to(DEV))[0].double().cpu()
        return ops.ConvFn.apply(a.to(DEV), w.to(DEV), None, one, pad, dl).double().cpu()

    def ref64(a, w):
        if transposed:
            return torch.nn.grad.conv3d_input((N, C, D, H, W), w.double(), a.double(), 1, pad, dl)
        return F.conv3d(a.double(), w.double(), None, 1, pad, dl)

    try:
        for positive in (False, True):
            a = rnd(N, K if transposed else C, D, H, W, seed=310)
            w = rnd(K, C, *ks, seed=311, scale=0.1)
            if positive:
                a, w = a.abs(), w.abs()
            if layout == 'wrows':                                              # the spread in the WEIGHTS: output row r at 2^-RANGE_EXPS[r % 5] (a scale per output row)
                rows = C if transposed else K
                sc = torch.pow(2.0, -torch.tensor(RANGE_EXPS, dtype=torch.float64)[torch.arange(rows) % 5])
                w = (w.double() * (sc.view(1, C, 1, 1, 1) if transposed else sc.view(K, 1, 1, 1, 1))).float()
            else:
                a = _spread(a, layout, seed=312)
            ref, den = ref64(a, w), ref64(a.abs(), w.abs())
            assert den.min().item() > 0

            def worst(path, guard=1):
                lib().call('dpf_set_f32_matrix_path', path)
                lib().call('dpf_debug_set_range_guard', guard)
                try:
                    return ((run(a, w) - ref).abs() / den).max().item()
                finally:
                    lib().call('dpf_debug_set_range_guard', 1)

            e0, e2 = worst(0), worst(2)
            assert e0 <= 5e-6, (positive, e0)                                 # (432 .. 1728 terms on the fp32 instruction)
            assert e2 <= 2 * e0 + 1e-7, (layout, positive, e2, e0)
            if layout in ('wbands', 'hbands'):
                eoff = worst(2, guard=0)
                # the regime the guard exists for is really probed here (eoff == e0: a forced chunk layout -- tests/test_gpu_fallbacks.py -- sent
                # this shape to the fp32 kernel, where there is no guard to switch off)
                assert eoff > 30 * e0 or eoff == e0, (layout, positive, eoff, e0)
    finally:
        lib().call('dpf_set_f32_matrix_path', prev)


@pytest.mark.parametrize('cfg', [(2, 32, 32, 8, 64, 128, (3, 3, 3), 1), (4, 64, 64, 1, 96, 192, (1, 3, 3), 1), (2, 48, 48, 1, 96, 192, (1, 3, 3), 2)])
@pytest.mark.parametrize('layout', ['wbands', 'elem'])
def test_conv_range_guard_is_run_to_run_bitwise_reproducible(cfg, layout):
    """The guarded convolution writes every output from one workgroup, without atomics: its result is a function of the data alone.  The
    deferral decisions travel through LDS flags between the waves of a workgroup (conv_igemm2.hip: lane_defers / s_red[8 + par]); a missed
    or stale flag would drop or repeat the deferred positions of a chunk in SOME launches.  Hundreds of workgroups with data that defers in
    most chunks, ten launches each of forward and data gradient: the same bits every time, and the same bits as the launch checked against
    fp64 (max error within the fp32-instruction bar of test_conv_f16_component_path_in_block_dynamic_range)."""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, dil = cfg
    pad = tuple(((k - 1) * dil) // 2 if k > 1 else 0 for k in ks)
    dl = tuple(dil if k > 1 else 1 for k in ks)
    one = (1, 1, 1)
    if lib().cdll.dpf_get_f32_matrix_path() != 2:
        pytest.skip('the range guard belongs to matrix path 2')
    x = _spread(rnd(N, C, D, H, W, seed=350), layout, seed=351).to(DEV)
    g = _spread(rnd(N, K, D, H, W, seed=352), layout, seed=353).to(DEV)
    w = rnd(K, C, *ks, seed=354, scale=0.1).to(DEV)

    def once():
        xin = x.clone().requires_grad_()
        y = ops.ConvFn.apply(xin, w, None, one, pad, dl)
        return y.detach(), torch.autograd.grad(y, xin, g)[0]

    y0, gx0 = once()
    ref = F.conv3d(x.double(), w.double(), None, 1, pad, dl)
    den = F.conv3d(x.double().abs(), w.double().abs(), None, 1, pad, dl)
    assert ((y0.double() - ref).abs() / den).max().item() <= 5e-6
    for i in range(10):
        y, gx = once()
        assert torch.equal(y, y0), ('forward', i, (y - y0).abs().max().item())
        assert torch.equal(gx, gx0), ('data gradient', i, (gx - gx0).abs().max().item())


@pytest.mark.parametrize('cfg', [(2, 32, 32, 4, 16, 64, (3, 3, 3), 1), (2, 24, 40, 1, 24, 96, (1, 3, 3), 3)])
@pytest.mark.parametrize('layout', ['chan', 'wbands', 'elem'])
@pytest.mark.parametrize('which', ['x', 'g', 'both'])
def test_weight_gradient_f16_component_path_in_block_dynamic_range(cfg, layout, which):
    """The weight gradient on dpf_set_f32_matrix_path(2) when one tile holds values at 1 .. 2^-34 of its maximum, in x, in g or in both
    (VERDICT r5 item 1a).  dW[k][c][t] sums over positions, so the spread that matters is the one ACROSS channels ('chan': channel c holds
    magnitude 2^-RANGE_EXPS[c % 5]): every g row and x channel carries its own exponent (conv_wgrad2.hip), and each dW element is compared
    with fp64 relative to sum |g| |x| over its own positions -- for all-positive data the element itself.  Bar: the worst element within
    2 x the worst element of the fp32 matrix instruction (path 0).  (Round 5's kernel -- one exponent per tile and operand -- misses the
    bar on the 'chan' layout by four orders of magnitude: profiles/r06_range_guard_before_after.txt.)"""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, dil = cfg
    pad = tuple(((k - 1) * dil) // 2 if k > 1 else 0 for k in ks)
    dl = tuple(dil if k > 1 else 1 for k in ks)
    st = (1, 1, 1)
    prev = lib().cdll.dpf_get_f32_matrix_path()
    try:
        for positive in (False, True):
            x, g = rnd(N, C, D, H, W, seed=320), rnd(N, K, D, H, W, seed=321)
            if positive:
                x, g = x.abs(), g.abs()
            if which in ('x', 'both'):
                x = _spread(x, layout, seed=322)
            if which in ('g', 'both'):
                g = _spread(g, layout, seed=323)
            ref = torch.nn.grad.conv3d_weight(x.double(), (K, C) + ks, g.double(), st, pad, dl)
            den = torch.nn.grad.conv3d_weight(x.abs().double(), (K, C) + ks, g.abs().double(), st, pad, dl)
            assert den.min().item() > 0

            def worst(path):
                lib().call('dpf_set_f32_matrix_path', path)
                got = ops._conv_wgrad_raw(g.to(DEV), x.to(DEV), (K, C) + ks, st, pad, dl).double().cpu()
                return ((got - ref).abs() / den).max().item()

            e0, e2 = worst(0), worst(2)
            assert e0 <= 2e-5, (positive, e0)                                  # (tens of thousands of positions per element)
            assert e2 <= 2 * e0 + 1e-7, (layout, which, positive, e2, e0)
    finally:
        lib().call('dpf_set_f32_matrix_path', prev)


@pytest.mark.parametrize('cfg', [(2, 32, 32, 8, 64, 128, (3, 3, 3), 1), (4, 64, 64, 1, 96, 192, (1, 3, 3), 1)])
def test_weight_gradient_range_guard_is_bitwise_reproducible_in_deterministic_mode(cfg):
    """The per-channel exponents of the weight-gradient kernel are raised through LDS atomics and a flag (conv_wgrad2.hip: own_scan /
    refresh_exponents); when a refresh happens decides at which scale the following segments are split.  In deterministic mode (tiles
    committed in a fixed order) ten launches on data whose channels sit at 1 .. 2^-34 and whose magnitudes also vary along W return the
    same bits."""
    from dualpixelface_amd._lib import lib
    ops = _ops()
    N, C, K, D, H, W, ks, dil = cfg
    pad = tuple(((k - 1) * dil) // 2 if k > 1 else 0 for k in ks)
    dl = tuple(dil if k > 1 else 1 for k in ks)
    st = (1, 1, 1)
    if lib().cdll.dpf_get_f32_matrix_path() != 2:
        pytest.skip('the range guard belongs to matrix path 2')
    x = _spread(_spread(rnd(N, C, D, H, W, seed=360), 'chan', 0), 'wbands', 0).to(DEV)
    g = _spread(rnd(N, K, D, H, W, seed=361), 'elem', seed=362).to(DEV)
    with ops.deterministic_mode():
        first = ops._conv_wgrad_raw(g, x, (K, C) + ks, st, pad, dl).clone()
        for i in range(10):
            again = ops._conv_wgrad_raw(g, x, (K, C) + ks, st, pad, dl)
            assert torch.equal(again, first), (i, (again - first).abs().max().item())
    ref = torch.nn.grad.conv3d_weight(x.double(), (K, C) + ks, g.double(), st, pad, dl)
    den = torch.nn.grad.conv3d_weight(x.abs().double(), (K, C) + ks, g.abs().double(), st, pad, dl)
    assert ((first.double() - ref).abs() / den).max().item() <= 2e-5


@pytest.mark.parametrize('cfg', [(1, 35, 64, 4, 12, 36), (1, 64, 64, 4, 8, 36)])
def test_deform_conv_backward_f16_component_path_in_block_dynamic_range(cfg):
    """The deformable conv backward forms gcol = W^T go from two f16 components per operand (dpf_set_f32_matrix_path(2)).  A gcol element
    sums over the output channels of ONE voxel, so the range guard scales every voxel by its own largest |go|.  Output gradients whose
    voxels lie at 1 .. 2^-34 of the tile's maximum (bands of 6 columns): grad_offset[.., voxel] is linear in go[.., voxel], so its error is
    measured relative to (the voxel's band scale x the rms of the unscaled gradient); grad_input relative to the same backward pass of
    (|W|, |go|).  fp64 oracle; bar: the worst element within 2 x the worst element on the fp32 matrix instruction (path 0)."""
    from dualpixelface_amd._lib import lib
    from oracle import dcn3d
    ops = _ops()
    B, C, K, D, H, W = cfg
    x, off = rnd(B, C, D, H, W, seed=330), rnd(B, 81, D, H, W, seed=331, scale=0.7)
    wt, bs = rnd(K, C, 3, 3, 3, seed=332, scale=0.1), rnd(K, seed=333)
    go_plain = rnd(B, K, D, H, W, seed=334)
    go = _spread(go_plain, 'wbands', seed=335)
    band = (go.double().abs().sum(dim=1, keepdim=True) / go_plain.double().abs().sum(dim=1, keepdim=True))      # [B,1,D,H,W]: 2^-e of the voxel
    xd, od, wd, bd = x.double(), off.double(), wt.double(), bs.double()
    gi_ref, goff_ref, _, _ = dcn3d.deform_conv3d_backward(xd, od, wd, bd, go.double())
    gi_den = dcn3d.deform_conv3d_backward(xd, od, wd.abs(), bd, go.double().abs())[0]
    goff_plain = dcn3d.deform_conv3d_backward(xd, od, wd, bd, go_plain.double())[1]
    goff_den = band * goff_plain.pow(2).mean().sqrt()
    prev = lib().cdll.dpf_get_f32_matrix_path()
    errs = {}
    try:
        for path in (0, 2):
            lib().call('dpf_set_f32_matrix_path', path)
            xg, og, wg, bg = [t.to(DEV).requires_grad_() for t in (x, off, wt, bs)]
            y = ops.deform_conv3d(xg, og, wg, bg)
            gi, goff = torch.autograd.grad(y, (xg, og), go.to(DEV))
            nz = gi_den > 0
            errs[path] = (((gi.double().cpu() - gi_ref).abs()[nz] / gi_den[nz]).max().item(),
                          ((goff.double().cpu() - goff_ref).abs() / goff_den).max().item())
    finally:
        lib().call('dpf_set_f32_matrix_path', prev)
    # (grad_input is committed through a fixed-point scatter whose unit follows the tile's largest contribution: both paths share that floor,
    # which is why it is compared between the paths and not held to an absolute bound)
    assert errs[2][0] <= 2 * errs[0][0] + 1e-7, errs
    assert errs[2][1] <= 2 * errs[0][1] + 1e-7, errs
    assert errs[0][1] <= 5e-5, errs


@pytest.mark.parametrize('groups', [(2, 1), (1, 2), (2, 2), (2, 4), (4, 2)])
def test_dcn_compat_module_group_and_deformable_group(groups):
    """group / deformable_group > 1 through the drop-in for the reference's pybind module (dcn_compat.py: slices, single-group HIP launches per
    piece, sums where a conv group spans several offset groups -- deform_conv_cuda.cu:65-66,84-121, deform_im2col_cuda.cuh:222-232), forward and
    all four gradients against the oracle's grouped forward and its fp64 autograd (tests/test_oracle_dcn.py pins that to F.conv3d(groups))."""
    from oracle import dcn3d
    import dualpixelface_amd.dcn_compat as DCN
    group, dg = groups
    B, C, K, D, H, W = 2, 16, 8, 3, 6, 12
    x, off = rnd(B, C, D, H, W, seed=340), rnd(B, dg * 81, D, H, W, seed=341, scale=1.2)
    w, b = rnd(K, C // group, 3, 3, 3, seed=342, scale=0.1), rnd(K, seed=343)
    go = rnd(B, K, D, H, W, seed=344)
    xd, od, wd, bd = [t.double().requires_grad_() for t in (x, off, w, b)]
    y_ref = dcn3d.deform_conv3d_forward_grouped(xd, od, wd, bd, group=group, deformable_group=dg)
    g_ref = torch.autograd.grad(y_ref, (xd, od, wd, bd), go.double())
    ints = (3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, group, dg, 64)
    y = DCN.deform_conv_forward(x.to(DEV), w.to(DEV), b.to(DEV), off.to(DEV), *ints)
    close(y, y_ref, 1e-4, 'grouped dcn fwd')
    got = DCN.deform_conv_backward(x.to(DEV), w.to(DEV), b.to(DEV), off.to(DEV), go.to(DEV), *ints)
    for a, r, nm in zip(got, g_ref, ('grad_input', 'grad_offset', 'grad_weight', 'grad_bias')):
        close(a, r, 2e-4, 'grouped dcn ' + nm)


@pytest.mark.parametrize('shape', [(1, 8, 16, 4, 6, 16), (1, 5, 7, 3, 5, 6)])
def test_deform_conv_integer_offsets_and_the_validity_rule(shape):
    """Integer offsets put samples exactly on voxel centres, on the borders and on coordinate -1: deform_im2col_cuda.cuh:248 declares a
    sample outside the OPEN interval (-1, size) invalid -- no value and no coordinate derivative -- although its high corner would still
    sit inside the volume.  (The lean kernels read a zero-padded image, which alone would hand such a sample a derivative.)"""
    from oracle import dcn3d
    ops = _ops()
    B, C, K, D, H, W = shape
    x = rnd(B, C, D, H, W, seed=170)
    off = torch.randint(-2, 3, (B, 81, D, H, W), generator=torch.Generator().manual_seed(171)).float()
    wt = rnd(K, C, 3, 3, 3, seed=172, scale=0.1)
    bs = rnd(K, seed=173)
    y_ref = dcn3d.deform_conv3d_forward(x, off, wt, bs)
    go = rnd(*y_ref.shape, seed=174)
    gr = dcn3d.deform_conv3d_backward(x, off, wt, bs, go)
    xg, og, wg, bg = [t.to(DEV).requires_grad_() for t in (x, off, wt, bs)]
    y = ops.deform_conv3d(xg, og, wg, bg)
    close(y, y_ref, 1e-4, 'dcn fwd (integer offsets)')
    gg = torch.autograd.grad(y, (xg, og, wg, bg), go.to(DEV))
    for a, r, nm in zip(gg, gr, ('grad_input', 'grad_offset', 'grad_weight', 'grad_bias')):
        close(a, r, 2e-4, 'dcn %s (integer offsets)' % nm)


@pytest.mark.parametrize('gi', [None, 5, 17])
def test_deform_conv_collapsed_samples(gi):
    """Offsets that send every sample of a 4x2x32 tile to the same point: the weight mass of one grad_input cell is ~6900, far above the
    bound the packed fixed-point scatter assumes for its first pass -- it must measure that and repeat the pass (no wrap-around).  `gi`
    exercises grad_input for a channel prefix (odd counts split a packed channel pair)."""
    from oracle import dcn3d
    ops = _ops()
    B, C, K, D, H, W = 1, 20, 8, 4, 4, 64
    x = rnd(B, C, D, H, W, seed=75)
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing='ij')
    off = torch.zeros(B, 81, D, H, W)
    for t in range(27):
        ti, tj, tk = t // 9, (t // 3) % 3, t % 3
        off[0, 3 * t] = 1.3 - (zz - 1 + ti).float()
        off[0, 3 * t + 1] = (2 * (yy // 2)).float() + 0.4 - (yy - 1 + tj).float()
        off[0, 3 * t + 2] = (32 * (xx // 32)).float() + 16.3 - (xx - 1 + tk).float()
    off = off + 0.01 * rnd(*off.shape, seed=76)
    wt = rnd(K, C, 3, 3, 3, seed=77, scale=0.1)
    bs = rnd(K, seed=78)
    y_ref = dcn3d.deform_conv3d_forward(x.double(), off.double(), wt.double(), bs.double())          # fp64 oracle: ~6900 terms per cell
    go = rnd(*y_ref.shape, seed=79)
    gr = dcn3d.deform_conv3d_backward(x.double(), off.double(), wt.double(), bs.double(), go.double())
    xg, og, wg, bg = [t.to(DEV).requires_grad_() for t in (x, off, wt, bs)]
    y = ops.deform_conv3d(xg, og, wg, bg, gi_channels=gi)
    close(y, y_ref, 1e-4, 'dcn fwd (collapsed)')
    gg = torch.autograd.grad(y, (xg, og, wg, bg), go.to(DEV))
    gx_ref = gr[0].clone()
    if gi is not None:
        gx_ref[:, gi:] = 0
    close(gg[0], gx_ref, 1e-4, 'dcn grad_input (collapsed)')   # fp32 coordinate arithmetic (like the reference) against the fp64 oracle
    for a, r, nm in zip(gg[1:], gr[1:], ('grad_offset', 'grad_weight', 'grad_bias')):
        close(a, r, 2e-4, 'dcn ' + nm)


def test_anm_volume_and_sigmoid_mean():
    from oracle import recipe_state
    from oracle.stereodpnet import StereoDPNetOracle
    from dualpixelface_amd.recipe import synthetic_batch
    ops = _ops()
    B, C, L, h, w = 2, 32, 8, 6, 10
    orc = StereoDPNetOracle({}, training=True)
    cost = rnd(B, C, L, h, w, seed=80).requires_grad_()
    batch = synthetic_batch(B, 4 * h, 4 * w, seed=1)
    disp_full = torch.rand(B, 4 * h, 4 * w, generator=torch.Generator().manual_seed(81)) * 16 - 4
    vol_r = orc.anm_front(cost, disp_full, batch['K'], batch['abvalue'])
    go = rnd(*vol_r.shape, seed=82)
    (gr,) = torch.autograd.grad(vol_r, cost, go)
    cg = cost.detach().to(DEV).requires_grad_()
    vol, idx = ops.anm_volume(cg, disp_full.to(DEV), batch['K'].to(DEV), batch['abvalue'].to(DEV), orc.cfg.costrange, 4)
    assert torch.equal(idx.cpu().long(), orc.taps['anm_idx']), 'sampled level indices must be bit-exact'
    close_elem(vol[:, :C], vol_r[:, :C], 'anm gathered cost')
    close(vol[:, C:], vol_r[:, C:], 1e-4, 'anm xyz')
    (gg,) = torch.autograd.grad(vol, cg, go.to(DEV))
    close_elem(gg, gr, 'anm bwd')
    u = rnd(B * 4, 3, 8, 12, seed=83).requires_grad_()
    y_ref = torch.sigmoid(u).view(B, 4, 3, 8, 12).mean(1) * 2.0 - 1.0
    go = rnd(*y_ref.shape, seed=84)
    (gr,) = torch.autograd.grad(y_ref, u, go)
    ug = u.detach().to(DEV).requires_grad_()
    y = ops.sigmoid_mean(ug, B, 4)
    close_elem(y, y_ref, 'sigmoid_mean')
    (gg,) = torch.autograd.grad(y, ug, go.to(DEV))
    close_elem(gg, gr, 'sigmoid_mean bwd')


def test_anm_level_indices_full_size_ties_and_boundaries():
    """Index work at the headline size: the HIP level selection and the oracle get the SAME disparity at 4 x 256 x 384 (quarter resolution of
    4 x 1024 x 1536) and must agree bit for bit.  The field is random values mixed with every value at which the selected set or its
    order can change -- each level, each midpoint between two levels, their +-1 and +-2 ulp neighbours, values far outside the range.
    Where the k-th and (k+1)-th scores are EXACTLY equal, torch.topk's choice is backend defined; there the reference's CUDA rule (lowest
    index among the equal ones) is the stated behaviour (oracle topk_ties='cuda'), everywhere else the literal torch.topk call decides."""
    from oracle.stereodpnet import StereoDPNetOracle
    from dualpixelface_amd.recipe import synthetic_batch
    ops = _ops()
    B, C, L, h, w = 4, 4, 8, 256, 384
    orc = StereoDPNetOracle({}, training=True)
    cr = torch.tensor(orc.cfg.costrange, dtype=torch.float32)
    special = [cr, (cr[:-1] + cr[1:]) / 2, torch.tensor([-50.0, -1.25, 2.75, 40.0, 0.1, 1e-7, -1e-7])]
    special = torch.cat(special)
    inf = torch.tensor(float('inf'))
    up, dn = torch.nextafter(special, inf), torch.nextafter(special, -inf)
    special = torch.cat([special, up, dn, torch.nextafter(up, inf), torch.nextafter(dn, -inf)])    # +-1 and +-2 ulp neighbours
    gen = torch.Generator().manual_seed(90)
    q = torch.rand(B, h, w, generator=gen) * 5.0 - 2.0            # quarter-resolution disparity (what the selection sees)
    pick = torch.randint(0, special.numel(), (B, h, w), generator=gen)
    use = torch.rand(B, h, w, generator=gen) < 0.5
    q = torch.where(use, special[pick], q)
    disp_full = torch.rand(B, 4 * h, 4 * w, generator=gen) * 16 - 4
    disp_full[:, ::4, ::4] = q * 4.0                              # nearest x0.25 reads pixel (4y, 4x); the x 0.25 that follows is exact
    cost = rnd(B, C, L, h, w, seed=91)
    batch = synthetic_batch(B, 4 * h, 4 * w, seed=1)
    _, idx = ops.anm_volume(cost.to(DEV), disp_full.to(DEV), batch['K'].to(DEV), batch['abvalue'].to(DEV), orc.cfg.costrange, 4)
    idx = idx.cpu().long()
    orc.anm_front(cost, disp_full, batch['K'], batch['abvalue'])
    idx_torch = orc.taps['anm_idx']
    orc.topk_ties = 'cuda'
    orc.anm_front(cost, disp_full, batch['K'], batch['abvalue'])
    idx_cuda = orc.taps['anm_idx']
    score = 1.0 / ((cr.view(1, -1, 1, 1) - q.unsqueeze(1)).abs() + 1e-6)
    srt = torch.sort(score, dim=1, descending=True)[0]
    tie = srt[:, 3] == srt[:, 4]                                  # membership tie: the 4th and 5th best scores are equal
    assert tie.float().mean() > 0.02 and (~tie).float().mean() > 0.5, 'the field must exercise both cases'
    assert torch.equal(idx, idx_cuda), 'level indices must be bit-exact (ties: lowest index, the CUDA top-k rule)'
    ne = (idx != idx_torch).any(1)
    assert not (ne & ~tie).any(), 'away from exact ties the literal torch.topk call decides'


@pytest.mark.parametrize('mode', ['ones', 'bern'])
def test_losses_against_reference_fixture(mode, golden_dir):
    ops = _ops()
    g = np.load(golden_dir + '/loss.npz')
    t = lambda k: torch.from_numpy(g[mode + '_' + k]).to(DEV)
    pd, pn = t('pred_depth').requires_grad_(), t('pred_normal').requires_grad_()
    out = ops.stereo_losses(pd, pn[:, 0], t('disp'), t('normal'), t('mask'), [1.0, 0.7, 0.5], 1.0, 1.0)
    for i, k in enumerate(('smoothL1_loss', 'cosine_loss', 'final_loss')):
        close(out[i], torch.from_numpy(g[mode + '_' + k]), 1e-5, k)
    gpd, gpn = torch.autograd.grad(out[2], (pd, pn))
    close_elem(gpd, torch.from_numpy(g[mode + '_g_pred_depth']), 'd pred_depth')
    close_elem(gpn, torch.from_numpy(g[mode + '_g_pred_normal']), 'd pred_normal')


def test_confidence_weighted_loss_against_reference_fixture(golden_dir):
    """The confidence-weighted smooth-L1 branch (src/loss/depth/smoothL1.py:33-36: prediction and target both multiplied by batch['conf']),
    through the plugin's loss_selector -- values and gradients from the imported reference (tests/golden/make_golden.py gen_loss)."""
    from dualpixelface_amd import load_option
    from dualpixelface_amd.losses import loss_selector
    g = np.load(golden_dir + '/loss.npz')
    t = lambda k: torch.from_numpy(g['conf_' + k]).to(DEV)
    pd, pn = t('pred_depth').requires_grad_(), t('pred_normal').requires_grad_()
    batch = {k: t(k) for k in ('disp', 'normal', 'mask', 'conf')}
    batch['abvalue'] = torch.zeros(pd.shape[0], 2, device=DEV)
    res = loss_selector(load_option()).forward({'pred_depth': pd, 'pred_normal': pn}, batch)
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        close(res[k], torch.from_numpy(g['conf_' + k]), 1e-5, k)
    gpd, gpn = torch.autograd.grad(res['final_loss'], (pd, pn))
    close_elem(gpd, torch.from_numpy(g['conf_g_pred_depth']), 'd pred_depth')
    close_elem(gpn, torch.from_numpy(g['conf_g_pred_normal']), 'd pred_normal')


def test_adam_step():
    from oracle.stereodpnet import adam_step as adam_ref
    ops = _ops()
    n = 10007
    p, g = rnd(n, seed=90), rnd(n, seed=91, scale=0.01)
    pr, m, v = {'p': p.clone()}, {'p': torch.zeros(n)}, {'p': torch.zeros(n)}
    pg, mg, vg, gg = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), g.to(DEV)
    for step in (1, 2, 3):
        adam_ref(pr, {'p': g}, m, v, step)
        ops.adam_step(pg, gg, mg, vg, step, 1e-4)
    close_elem(pg, pr['p'], 'adam param')
    close_elem(mg, m['p'], 'adam m')
    close_elem(vg, v['p'], 'adam v')


def test_psm_volume_against_reference_fixture(golden_dir):
    ops = _ops()
    g = np.load(golden_dir + '/psmnet_volume.npz')
    ref, tar = torch.from_numpy(g['ref']).to(DEV), torch.from_numpy(g['tar']).to(DEV)
    costrange = [i * 0.5 - 1.0 for i in range(8)]
    shifts = [int(c) for c in costrange]                      # psmnet/modules.py:229 (truncation toward zero)
    assert shifts == [-1, 0, 0, 0, 1, 1, 2, 2]
    close(ops.psm_volume(ref, tar, shifts, 0), torch.from_numpy(g['vol_psmnet']), 0.0, 'psmnet volume')
    close_elem(ops.psm_volume(ref, tar, shifts, 40), torch.from_numpy(g['vol_gwcnet']), 'gwcnet volume')


def test_psm_volume_full_size_bit_exact():
    """BASELINE configs[3] at its stated size (c4 was pinned at the 256 x 256 fixture only): the PSMNet cost-volume path on
    2 x 1024 x 1536 pairs = 32-channel features at 256 x 384, 8 integer shifts.  The oracle's volume is pure indexing, so it runs on the
    device tensors as well: the concat volume must be BIT-EXACT (403 MB), the group-wise correlation channels and the backward element-wise
    close."""
    from oracle.psmnet_volume import psm_volume
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    ref = torch.randn(2, 32, 256, 384, generator=g).to(DEV)
    tar = torch.randn(2, 32, 256, 384, generator=g).to(DEV)
    costrange = [i * 0.5 - 1.0 for i in range(8)]
    shifts = [int(c) for c in costrange]
    vol = ops.psm_volume(ref, tar, shifts, 0)
    want = psm_volume(ref, tar, costrange, 0)
    assert vol.shape == want.shape == (2, 64, 8, 256, 384) and torch.equal(vol, want)
    del vol, want
    rg, tg = ref.clone().requires_grad_(), tar.clone().requires_grad_()
    v2 = ops.psm_volume(rg, tg, shifts, 8)
    ro, to = ref.clone().requires_grad_(), tar.clone().requires_grad_()
    w2 = psm_volume(ro, to, costrange, 8)
    assert torch.equal(v2[:, :64], w2[:, :64])
    close_elem(v2[:, 64:], w2[:, 64:].detach().cpu(), 'gwc channels at full size')
    go = torch.randn(v2.shape, generator=torch.Generator().manual_seed(32)).to(DEV)
    g1 = torch.autograd.grad(v2, (rg, tg), go)
    g2 = torch.autograd.grad(w2, (ro, to), go)
    for a, b, nm in zip(g1, g2, ('d ref', 'd tar')):
        close_elem(a, b.cpu(), 'psm volume backward ' + nm)


def test_layout_kernels():
    ops = _ops()
    a, b, c = rnd(2, 5, 3, 7, seed=100).requires_grad_(), rnd(2, 8, 3, 7, seed=101).requires_grad_(), rnd(2, 1, 3, 7, seed=102).requires_grad_()
    ref = torch.cat([a, b, c], 1)
    go = rnd(*ref.shape, seed=103)
    gr = torch.autograd.grad(ref, (a, b, c), go)
    ag, bg, cg = [t.detach().to(DEV).requires_grad_() for t in (a, b, c)]
    out = ops.concat_channels([ag, bg, cg])
    close(out, ref, 0.0, 'concat')
    for g1, g2 in zip(torch.autograd.grad(out, (ag, bg, cg), go.to(DEV)), gr):
        close(g1, g2, 0.0, 'concat bwd')
    xs = [rnd(3, 4, 6, seed=110 + i) for i in range(3)]
    close(ops.stack_dim1([t.to(DEV) for t in xs]), torch.stack(xs, 1), 0.0, 'stack')
    x = rnd(2, 6, 4, 5, 3, seed=120).requires_grad_()
    ref = x.permute(0, 2, 1, 3, 4).contiguous()
    go = rnd(*ref.shape, seed=121)
    (gr,) = torch.autograd.grad(ref, x, go)
    xg = x.detach().to(DEV).requires_grad_()
    y = ops.swap_axes12(xg)
    close(y, ref, 0.0, 'swap axes')
    (gg,) = torch.autograd.grad(y, xg, go.to(DEV))
    close(gg, gr, 0.0, 'swap axes bwd')
    close(ops.channel_max(x.detach().to(DEV)), x.detach().max(1)[0], 0.0, 'channel max')
    r, af, ab = rnd(32, seed=130), rnd(32, seed=131), rnd(32, seed=132)
    rg = r.to(DEV)
    ops.bn_replay(rg, af.to(DEV), ab.to(DEV), 0.9 ** 16, 0.9 * 4.2, 4.2)
    close(rg, r * 0.9 ** 16 + af * (0.9 * 4.2) + ab * 4.2, 1e-6, 'bn replay')


BF16_CASES = [
    # N, C, H, W, K, k, stride, pad, dil
    (2, 32, 20, 45, 32, 3, 1, 1, 1),
    (1, 3, 33, 70, 32, 3, 2, 1, 1),       # first conv: 3 channels in one 16-wide chunk, stride 2
    (2, 96, 17, 35, 32, 3, 1, 5, 5),      # dilated
    (1, 64, 12, 40, 192, 1, 1, 0, 1),     # 1x1, > 128 output channels (two launches)
    (2, 40, 16, 24, 64, 3, 2, 2, 2),      # stride 2 + dilation: data gradient falls back to the fp32 kernel
]


@pytest.mark.parametrize('case', BF16_CASES)
def test_conv2d_bf16_operands(case):
    """bf16-operand conv == fp32 conv of the bf16-rounded operands (products are exact in fp32; only the summation order
    differs): 1e-5 of the tensor scale.  The weight gradient runs on the fp32 kernel with unrounded operands."""
    ops = _ops()
    N, C, H, W, K, k, st, pd, dl = case
    x = rnd(N, C, H, W, seed=90).requires_grad_()
    w = rnd(K, C, k, k, seed=91, scale=0.1).requires_grad_()
    b = rnd(K, seed=92).requires_grad_()
    xr, wr = x.detach().bfloat16().float().requires_grad_(), w.detach().bfloat16().float().requires_grad_()
    y_ref = F.conv2d(xr, wr, b, st, pd, dl)
    go = rnd(*y_ref.shape, seed=93)
    xg, wg, bg = [t.detach().to(DEV).requires_grad_() for t in (x, w, b)]
    y = ops.conv2d(xg, wg, bg, st, pd, dl, bf16=True)
    close(y, y_ref, 1e-5, 'bf16 conv fwd')
    gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), go.to(DEV))
    if st == 1:
        (gx_r,) = torch.autograd.grad(F.conv2d(xr, wr, None, st, pd, dl), xr, go.bfloat16().float())   # bf16(go) * bf16(w)
        close(gx, gx_r, 1e-5, 'bf16 conv dgrad')
    else:
        (gx_r,) = torch.autograd.grad(F.conv2d(x, w, None, st, pd, dl), x, go)
        close(gx, gx_r, 1e-4, 'fp32 fallback dgrad')
    (gw_r,) = torch.autograd.grad(F.conv2d(x, w, None, st, pd, dl), w, go)
    close(gw, gw_r, 2e-4, 'fp32 wgrad')
    close(gb, go.sum((0, 2, 3)), 1e-4, 'bgrad')


BF16_OPERAND_CASES = [c for c in CONV_CASES if c[4] % 4 == 0 and c[6] != (1, 1, 1) and c[5] > 4]


@pytest.mark.parametrize('case', BF16_OPERAND_CASES)
def test_conv_operands_bf16(case):
    """ops.conv_operands(True) (precision 16 / 'bf16'): the LDS-DMA kernels round their operands to bf16 while staging them.  A launch must
    equal an fp32 convolution of the bf16-rounded operands up to summation order (products of bf16 values are exact in fp32) -- 1e-5 of
    the tensor scale for forward / data gradient, 2e-4 for the weight gradient -- or, for the launches the bf16 kernels do not cover
    (8-channel stride-2 patches do not fit the LDS; > 128 output rows; the class-fused stride-2 data gradient), the exact fp32 result.
    The 3x3x3 stride-1 cases must take the bf16 kernels in all three directions."""
    ops = _ops()
    N, C, D, H, W, K, ks, st, pd, dl = case
    x = rnd(N, C, D, H, W, seed=1)
    w = rnd(K, C, *ks, seed=2, scale=0.1)
    b = rnd(K, seed=3)
    rb = lambda t: t.bfloat16().float()
    xg, wg, bg = [t.to(DEV).requires_grad_() for t in (x, w, b)]
    with ops.conv_operands(True):
        y = ops.ConvFn.apply(xg, wg, bg, st, pd, dl)
    assert not ops.CONV_OPERANDS_BF16
    go = rnd(*y.shape, seed=4)
    gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), go.to(DEV))          # outside the context: the node re-enters its precision

    def refs(xx, ww, gg):
        xx, ww = xx.clone().requires_grad_(), ww.clone().requires_grad_()
        yy = F.conv3d(xx, ww, b, st, pd, dl)
        return (yy.detach(),) + torch.autograd.grad(yy, (xx, ww), gg)

    y_b, gx_b, gw_b = refs(rb(x), rb(w), rb(go))
    y_f, gx_f, gw_f = refs(x, w, go)
    must = ks == (3, 3, 3) and st == (1, 1, 1) and K <= 128 and C <= 128
    for name, got, ref_b, ref_f, tol in (('fwd', y, y_b, y_f, 1e-5), ('dgrad', gx, gx_b, gx_f, 1e-5), ('wgrad', gw, gw_b, gw_f, 2e-4)):
        scale = float(ref_f.abs().max())
        # one launch covers <= 128 output channels and picks its kernel on its own (27 x 128 bf16 weight rows + the patch exceed the LDS)
        step = 128 if name != 'wgrad' else got.shape[0]
        for c0 in range(0, got.shape[1] if name != 'wgrad' else 1, step):
            sl = (slice(None), slice(c0, c0 + step)) if name != 'wgrad' else (slice(None),)
            eb, ef = float((got.cpu()[sl] - ref_b[sl]).abs().max()) / scale, float((got.cpu()[sl] - ref_f[sl]).abs().max()) / scale
            assert min(eb, ef) < tol, (name, c0, eb, ef)
            # the class-fused stride-2 data gradient (16-byte aligned g rows) has its bf16 variant too
            s2 = name == 'dgrad' and ks == (3, 3, 3) and st == (2, 2, 2) and y.shape[4] % 4 == 0 and K <= 128
            if must or s2:
                assert eb < ef and eb < tol, (name, 'expected the bf16-operand kernel', eb, ef)
    close(gb, go.sum((0, 2, 3, 4)), 1e-4, 'bgrad')
    # and the default precision is untouched afterwards
    y32 = ops.ConvFn.apply(xg, wg, bg, st, pd, dl)
    close(y32, y_f, 1e-4, 'fp32 after the context')


def test_avg_pool_and_resize_and_psm_volume_backward():
    """PSMNet-specific operators: AvgPool2d(k, k), bilinear resize to an arbitrary size (align_corners=True, incl. from 1x1), and the
    gradient of the integer-shift volume (concat and group-wise correlation) against autograd through the oracle."""
    from oracle.psmnet_volume import psm_volume
    ops = _ops()
    x = rnd(2, 8, 19, 33, seed=100).requires_grad_()
    for k in (8, 4, 16):
        y_ref = F.avg_pool2d(x, (k, k), (k, k))
        go = rnd(*y_ref.shape, seed=101)
        (gx_r,) = torch.autograd.grad(y_ref, x, go)
        xg = x.detach().to(DEV).requires_grad_()
        y = ops.avg_pool2d(xg, k)
        close_elem(y, y_ref, 'avg_pool fwd k=%d' % k)
        (gx,) = torch.autograd.grad(y, xg, go.to(DEV))
        close_elem(gx, gx_r, 'avg_pool bwd k=%d' % k)
    for shape in ((1, 1), (2, 3), (8, 5)):
        s = rnd(2, 8, *shape, seed=102).requires_grad_()
        y_ref = F.interpolate(s, size=(16, 24), mode='bilinear', align_corners=True)
        go = rnd(*y_ref.shape, seed=103)
        (gs_r,) = torch.autograd.grad(y_ref, s, go)
        sg = s.detach().to(DEV).requires_grad_()
        y = ops.resize_bilinear(sg, 16, 24)
        close_elem(y, y_ref, 'resize fwd %s' % (shape,))
        (gs,) = torch.autograd.grad(y, sg, go.to(DEV))
        close_elem(gs, gs_r, 'resize bwd %s' % (shape,))
    ref, tar = rnd(2, 40, 12, 9, seed=104).requires_grad_(), rnd(2, 40, 12, 9, seed=105).requires_grad_()
    costrange = [i * 0.5 - 1.0 for i in range(8)]                 # int() -> -1, 0, 0, 0, 0, 1, 1, 1 ... mixed signs
    costrange[0], costrange[7] = -2.0, 3.0
    for groups in (0, 40, 8):
        v_ref = psm_volume(ref, tar, costrange, groups)
        go = rnd(*v_ref.shape, seed=106)
        gr_r, gt_r = torch.autograd.grad(v_ref, (ref, tar), go)
        rg, tg = ref.detach().to(DEV).requires_grad_(), tar.detach().to(DEV).requires_grad_()
        v = ops.psm_volume(rg, tg, [int(d) for d in costrange], groups)
        close_elem(v, v_ref, 'psm volume g=%d' % groups)
        gr, gt = torch.autograd.grad(v, (rg, tg), go.to(DEV))
        close_elem(gr, gr_r, 'psm dref g=%d' % groups)
        close_elem(gt, gt_r, 'psm dtar g=%d' % groups)


def test_full_size_properties():
    """BASELINE-size tensors (4 x 32 x 8 x 256 x 384 aggregation conv, 4 x 35 x 4 x 256 x 384 deformable conv) through properties that
    need no CPU reference: linearity of the forward, the adjoint identities <conv(x), g> = <x, dgrad(g)> = <w, wgrad(x, g)> that tie the
    three conv kernels together, and the deformable conv's known answer (zero offsets == plain conv3d) plus its adjoints."""
    ops = _ops()
    gen = torch.Generator(device=DEV).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    # fp32 outputs carry ~1e-6 relative noise per element, so inner products agree to ~1e-6 of ||out|| * ||g|| (Cauchy-Schwarz
    # scale), not of their (cancelling) value
    scale = lambda a, b: (dot(a, a) * dot(b, b)) ** 0.5
    # ---- dense 3-D conv at the hourglass shape
    x, y = r(4, 32, 8, 256, 384), r(4, 32, 8, 256, 384)
    w = (r(32, 32, 3, 3, 3) * 0.05).requires_grad_()
    xg = x.clone().requires_grad_()
    out = ops.conv3d(xg, w, None, 1, 1, 1)
    lin = ops.conv3d(0.5 * x - 2.0 * y, w, None, 1, 1, 1) - (0.5 * out.detach() - 2.0 * ops.conv3d(y, w, None, 1, 1, 1).detach())
    assert float(lin.abs().max()) <= 2e-4 * float(out.detach().abs().max())
    g = r(*out.shape)
    gx, gw = torch.autograd.grad(out, (xg, w), g)
    a0, a1, a2 = dot(out.detach(), g), dot(x, gx), dot(w.detach(), gw)
    tol = 2e-6 * scale(out.detach(), g)
    assert abs(a0 - a1) <= tol and abs(a0 - a2) <= tol, (a0, a1, a2, tol)
    del x, y, out, lin, g, gx, xg
    # ---- stride-2 conv and its transposed data gradient
    x2 = r(4, 32, 8, 256, 384).requires_grad_()
    w2 = (r(64, 32, 3, 3, 3) * 0.05).requires_grad_()
    o2 = ops.conv3d(x2, w2, None, 2, 1, 1)
    g2 = r(*o2.shape)
    gx2, gw2 = torch.autograd.grad(o2, (x2, w2), g2)
    b0, tol = dot(o2.detach(), g2), 2e-6 * scale(o2.detach(), g2)
    assert abs(b0 - dot(x2.detach(), gx2)) <= tol and abs(b0 - dot(w2.detach(), gw2)) <= tol
    del x2, o2, g2, gx2
    # ---- deformable conv at the ANM shape: zero offsets == conv3d; adjoints with random offsets (beyond the halo too)
    xd = r(4, 35, 4, 256, 384).requires_grad_()
    wd = (r(64, 35, 3, 3, 3) * 0.05).requires_grad_()
    bd = r(64).requires_grad_()
    z = torch.zeros(4, 81, 4, 256, 384, device=DEV)
    y0 = ops.deform_conv3d(xd, z, wd, bd)
    yc = ops.conv3d(xd, wd, bd, 1, 1, 1)
    assert float((y0 - yc).abs().max()) <= 1e-4 * float(yc.abs().max())
    off = (r(4, 81, 4, 256, 384) * 1.5).requires_grad_()
    yd = ops.deform_conv3d(xd, off, wd, bd)
    gd = r(*yd.shape)
    gxd, god, gwd, gbd = torch.autograd.grad(yd, (xd, off, wd, bd), gd)
    c0, tol = dot(yd.detach(), gd), 2e-6 * scale(yd.detach(), gd)
    # y is linear in (x), in (w, b) jointly: <y, g> = <x, dx> + <b, db> = <w, dw> + <b, db>
    cb = dot(bd.detach(), gbd)
    assert abs(c0 - (dot(xd.detach(), gxd) + cb)) <= tol, (c0, dot(xd.detach(), gxd), cb, tol)
    assert abs(c0 - (dot(wd.detach(), gwd) + cb)) <= tol, (c0, dot(wd.detach(), gwd), cb, tol)
    # grad_offset: directional derivative along a random direction by central differences (in float32: coarse tolerance)
    d = r(*off.shape)
    eps = 1e-2
    with torch.no_grad():
        yp = ops.deform_conv3d(xd.detach(), off.detach() + eps * d, wd.detach(), bd.detach())
        ym = ops.deform_conv3d(xd.detach(), off.detach() - eps * d, wd.detach(), bd.detach())
    fd = dot(yp - ym, gd) / (2 * eps)
    an = dot(god, d)
    assert abs(fd - an) <= 5e-2 * abs(an) + 50 * tol, (fd, an, tol)


def test_full_size_properties_2d_head_norm_bf16():
    """More BASELINE-size launches through reference-free properties: the dilated 2-D conv of the normal head's stack
    (16 x 96 x 256 x 384, dilation 2: adjoints), the soft-argmin head (4 x 1 x 8 x 256 x 384 -> 1024 x 1536: the probabilities sum to 1,
    the prediction is their expectation, the gradient matches a central difference), BatchNorm forward / backward at the hourglass
    shape (4 x 32 x 8 x 256 x 384: zero mean, unit variance, sum dx = sum dx * xhat = 0 per channel), and the bf16-operand 3x3x3 kernel,
    which must equal the exact-fp32 kernel run on bf16-rounded operands (same accumulation type, only the summation order differs)."""
    ops = _ops()
    gen = torch.Generator(device=DEV).manual_seed(6)
    r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    scale = lambda a, b: (dot(a, a) * dot(b, b)) ** 0.5
    # ---- dilated 2-D conv (normal head, n_convs.1: 96 -> 96, dilation 2)
    x = r(16, 96, 256, 384).requires_grad_()
    w = (r(96, 96, 3, 3) * 0.03).requires_grad_()
    out = ops.conv2d(x, w, None, 1, 2, 2)
    g = r(*out.shape)
    gx, gw = torch.autograd.grad(out, (x, w), g)
    a0, tol = dot(out.detach(), g), 2e-6 * scale(out.detach(), g)
    assert abs(a0 - dot(x.detach(), gx)) <= tol and abs(a0 - dot(w.detach(), gw)) <= tol, (a0, dot(x.detach(), gx), dot(w.detach(), gw), tol)
    del x, out, g, gx
    # ---- soft-argmin head
    disp = [-4 + 0.5 * i for i in range(32)]
    lg = (r(4, 1, 8, 256, 384) * 2.0).requires_grad_()
    pred, prob = ops.softargmin(lg, disp, 4, True)
    assert pred.shape == (4, 1024, 1536) and prob.shape == (4, 32, 1024, 1536)
    ps = prob.sum(1)
    assert float((ps - 1).abs().max()) <= 1e-5
    ex = (prob * torch.tensor(disp, device=DEV).view(1, -1, 1, 1)).sum(1)
    assert float((ex - pred.detach()).abs().max()) <= 1e-4
    assert float(pred.detach().min()) >= -4.0 - 1e-4 and float(pred.detach().max()) <= 11.5 + 1e-4
    gp = r(*pred.shape)
    (gl,) = torch.autograd.grad(pred, lg, gp)
    d = r(*lg.shape)
    eps = 1e-2
    with torch.no_grad():
        pp = ops.softargmin(lg.detach() + eps * d, disp, 4, False)[0]
        pm = ops.softargmin(lg.detach() - eps * d, disp, 4, False)[0]
    fd, an = dot(pp - pm, gp) / (2 * eps), dot(gl, d)
    assert abs(fd - an) <= 2e-2 * abs(an) + 1e-4 * scale(gl, d), (fd, an)
    del prob, ps, ex, pred, gp, gl, pp, pm
    # ---- BatchNorm (training) forward / backward
    xb = (r(4, 32, 8, 256, 384) * 3.0 + 1.5).requires_grad_()
    wb, bb = torch.ones(32, device=DEV, requires_grad=True), torch.zeros(32, device=DEV, requires_grad=True)
    rm, rv = torch.zeros(32, device=DEV), torch.ones(32, device=DEV)
    yb = ops.norm_act(xb, wb, bb, running_mean=rm, running_var=rv, mode=1, act=0)
    m = yb.detach().double().mean((0, 2, 3, 4))
    v = yb.detach().double().var((0, 2, 3, 4), unbiased=False)
    assert float(m.abs().max()) <= 1e-5 and float((v - 1).abs().max()) <= 1e-4, (float(m.abs().max()), float((v - 1).abs().max()))
    assert float((rm - 0.1 * xb.detach().double().mean((0, 2, 3, 4)).float()).abs().max()) <= 1e-5
    gb_ = r(*yb.shape)
    dxb, dwb, dbb = torch.autograd.grad(yb, (xb, wb, bb), gb_)
    n = xb.numel() // 32
    s1 = dxb.double().sum((0, 2, 3, 4)).abs().max()
    s2 = (dxb.double() * yb.detach().double()).sum((0, 2, 3, 4)).abs().max()
    bound = float(dxb.double().abs().sum((0, 2, 3, 4)).max())
    assert float(s1) <= 2e-5 * bound and float(s2) <= 2e-5 * bound * 3, (float(s1), float(s2), bound)
    assert float((dbb.double() - gb_.double().sum((0, 2, 3, 4))).abs().max()) <= 1e-5 * float(gb_.double().abs().sum((0, 2, 3, 4)).max())
    assert float((dwb.double() - (gb_.double() * yb.detach().double()).sum((0, 2, 3, 4))).abs().max()) <= 1e-5 * float(gb_.double().abs().sum((0, 2, 3, 4)).max()) * 4
    del yb, gb_, dxb
    # ---- bf16-operand 3x3x3 kernel == exact kernel on bf16-rounded operands (sample 0)
    xc = xb.detach()
    wc = r(32, 32, 3, 3, 3) * 0.05
    with ops.conv_operands(True):
        ob = ops.conv3d(xc, wc, None, 1, 1, 1)
    oe = ops.conv3d(xc[:1].bfloat16().float(), wc.bfloat16().float(), None, 1, 1, 1)
    assert float((ob[:1] - oe).abs().max()) <= 1e-5 * float(oe.abs().max()), float((ob[:1] - oe).abs().max())
    of = ops.conv3d(xc[:1], wc, None, 1, 1, 1)
    assert float((ob[:1] - of).abs().max()) > 1e-4 * float(of.abs().max())        # and it is NOT the exact kernel (the mode engaged)


@pytest.mark.parametrize('shape', [
    # N, C, K, D, H, W, kernel, stride, bias
    (2, 8, 32, 1, 20, 36, (1, 3, 3), 1, False),          # ragged tiles in H and W
    (1, 16, 81, 3, 9, 40, (3, 3, 3), 1, False),          # three row tiles, last one partial
    (2, 32, 64, 4, 16, 32, (3, 3, 3), 2, False),         # stride 2
    (3, 12, 35, 1, 33, 68, (1, 3, 3), 1, True),          # bias is part of the normalised value
])
def test_conv_epilogue_batchnorm_statistics(shape):
    """dpf_conv_forward_stats + dpf_bn_finalize_partials (BatchNorm statistics from the conv epilogue) against the separate
    statistics pass and against torch's batch_norm on the CPU; running statistics included; bitwise reproducible."""
    ops = _ops()
    N, C, K, D, H, W, ks, stride, has_bias = shape
    x = rnd(N, C, D, H, W, seed=70)
    w = rnd(K, C, *ks, seed=71) * 0.2
    b = rnd(K, seed=72) if has_bias else None
    gamma, beta = rnd(K, seed=73).abs() + 0.5, rnd(K, seed=74)
    pad = tuple(k // 2 for k in ks)
    y_ref = F.conv3d(x, w, b, stride, pad)
    rm_ref, rv_ref = torch.zeros(K), torch.ones(K)
    z_ref = F.relu(F.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    outs = []
    for fuse in (True, False, True):
        ops.FUSE_BN_STATS = fuse
        st = {}
        rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        y = ops.conv3d(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), stride, pad, 1, stats=st)
        assert bool(st) == fuse, 'the LDS-DMA kernel should have taken this shape'
        z = ops.norm_act(y, gamma.to(DEV), beta.to(DEV), None, None, None, rm, rv, 1, 1, stats=st)
        assert not st                                                   # consumed
        outs.append((z, rm, rv))
        close(z, z_ref, 1e-4, 'bn(conv) fuse=%s' % fuse)
        close(rm, rm_ref, 1e-5, 'running_mean')
        close(rv, rv_ref, 1e-5, 'running_var')
    ops.FUSE_BN_STATS = True
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]) and torch.equal(outs[0][2], outs[2][2])
    # a stale holder (different tensor) must be ignored, not trusted
    st = {'ptr': 1234, 'channels': K, 'count': 1, 'slab': None, 'parts': 0}
    rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
    z = ops.norm_act(outs[0][0].new_tensor(y_ref.numpy()), gamma.to(DEV), beta.to(DEV), None, None, None, rm, rv, 1, 1, stats=st)
    close(z, z_ref, 1e-4, 'stale holder ignored')


@pytest.mark.parametrize('shape', [(2, 3, 1, 1, 16, 24), (1, 5, 2, 3, 16, 24), (2, 4, 4, 6, 16, 24), (1, 2, 8, 12, 5, 7), (1, 2, 7, 9, 28, 36)])
def test_resize_bilinear_half_pixel(shape):
    """F.interpolate(size, mode='bilinear', align_corners=False) (nnet/modules.py:110-120) and its adjoint, incl. 1x1 sources,
    downsampling and non-integer ratios."""
    ops = _ops()
    N, C, h, w, H, W = shape
    x = rnd(N, C, h, w, seed=80).requires_grad_()
    ref = F.interpolate(x, size=(H, W), mode='bilinear', align_corners=False)
    go = rnd(*ref.shape, seed=81)
    (gr,) = torch.autograd.grad(ref, x, go)
    xg = x.detach().to(DEV).requires_grad_()
    out = ops.resize_bilinear(xg, H, W, align_corners=False)
    close_elem(out, ref, 'half-pixel resize fwd')
    (gg,) = torch.autograd.grad(out, xg, go.to(DEV))
    close_elem(gg, gr, 'half-pixel resize bwd')


def test_l2_normalize_and_xyz_volume():
    ops = _ops()
    x = rnd(2, 3, 9, 14, seed=82)
    x[0, :, 0, 0] = 0.0                                              # zero vector: eps branch
    x = x.requires_grad_()
    ref = F.normalize(x, dim=1)
    go = rnd(*ref.shape, seed=83)
    (gr,) = torch.autograd.grad(ref, x, go)
    xg = x.detach().to(DEV).requires_grad_()
    out = ops.l2_normalize(xg)
    close_elem(out, ref, 'normalize fwd')
    (gg,) = torch.autograd.grad(out, xg, go.to(DEV))
    mask = torch.ones_like(gr)
    mask[0, :, 0, 0] = 0                                             # at the zero vector torch's gradient is g / eps = 1e12 * g: skip
    close(gg.cpu() * mask, gr * mask, 1e-5, 'normalize bwd')
    # coordinate volume at an arbitrary channel offset against the oracle's grid_maker_3d
    from oracle.nnet import NNetOracle
    from oracle.stereodpnet import Cfg
    from dualpixelface_amd.recipe import synthetic_batch
    batch = synthetic_batch(2, 32, 48, seed=3)
    B, L, h, w = 2, 8, 8, 12
    cfg = Cfg()
    levels = torch.tensor(cfg.costrange, dtype=torch.float32).view(1, L, 1, 1).expand(B, L, h, w).contiguous()
    vol = torch.zeros(B, 5, L, h, w, device=DEV)
    ops.xyz_volume_into(vol, 1, levels.to(DEV), batch['K'].to(DEV), batch['abvalue'].to(DEV))
    orc = NNetOracle({}, cfg, training=False)
    # reuse the oracle's arithmetic through its taps: run only the front of normal_module by feeding zero features
    class _Stop(Exception):
        pass
    def stop(*a, **k):
        raise _Stop()
    orc.convbn3 = stop
    try:
        orc.normal_module(torch.zeros(B, 2, L, h, w), {'K': batch['K'], 'abvalue': batch['abvalue']})
    except _Stop:
        pass
    close(vol[:, 1:4], orc.taps['wc'][:, :3], 1e-5, 'xyz volume')
    assert float(vol[:, 0].abs().max()) == 0 and float(vol[:, 4].abs().max()) == 0


@pytest.mark.parametrize('dims', [(2, 8, 6, 10), (1, 8, 16, 40), (1, 5, 7, 9)])
def test_softargmin_head_half_pixel(dims):
    """x4 trilinear (align_corners=False) + softmax + expectation (nnet/mainmodel.py:150-153, modules.py:196-217) and its gradient."""
    ops = _ops()
    B, D, h, w = dims
    k = rnd(B, 1, D, h, w, seed=90).requires_grad_()
    L = 4 * D
    disp = [i * (16.0 / L) - 4.0 for i in range(L)]
    up = F.interpolate(k, scale_factor=4, mode='trilinear', align_corners=False).squeeze(1)
    pr = F.softmax(up, 1)
    ref = torch.sum(pr * torch.tensor(disp).view(1, L, 1, 1), 1)
    go = rnd(*ref.shape, seed=91)
    (gr,) = torch.autograd.grad(ref, k, go)
    kg = k.detach().to(DEV).requires_grad_()
    pred, prob = ops.softargmin(kg, disp, 4, True, align_corners=False)
    close(pred, ref, 1e-5, 'head fwd')
    close(prob, pr, 1e-5, 'head prob')
    (gg,) = torch.autograd.grad(pred, kg, go.to(DEV))
    close(gg, gr, 1e-4, 'head bwd')


def test_diff_volume():
    """StereoNet's difference volume (stereonet/mainmodel.py:97-112) and its adjoint, positive / zero / negative shifts."""
    ops = _ops()
    B, C, h, w = 2, 5, 9, 14
    shifts = [-2, -1, 0, 0, 1, 3]
    ref = rnd(B, C, h, w, seed=95).requires_grad_()
    tar = rnd(B, C, h, w, seed=96).requires_grad_()
    parts = []
    for d in shifts:
        lvl = torch.zeros(B, C, h, w)
        if d == 0:
            lvl = ref - tar
        elif d > 0:
            lvl = torch.cat([ref[:, :, :-d] - tar[:, :, d:], lvl[:, :, h - d:]], 2)
        else:
            lvl = torch.cat([lvl[:, :, :-d], ref[:, :, -d:] - tar[:, :, :d]], 2)
        parts.append(lvl)
    want = torch.stack(parts, 2)
    go = rnd(*want.shape, seed=97)
    gr = torch.autograd.grad(want, (ref, tar), go)
    rg, tg = ref.detach().to(DEV).requires_grad_(), tar.detach().to(DEV).requires_grad_()
    vol = ops.diff_volume(rg, tg, shifts)
    assert torch.equal(vol.cpu(), want.detach())
    gg = torch.autograd.grad(vol, (rg, tg), go.to(DEV))
    close_elem(gg[0], gr[0], 'diff volume dref')
    close_elem(gg[1], gr[1], 'diff volume dtar')


def test_norm_act_concat_matches_separate_norm_and_cat():
    """BatchNorm branches written straight into their slice of the concatenation (dpf_norm_act_*_slice) against
    torch.cat([batch_norm(x_i)]) with autograd, training and eval, running statistics included."""
    ops = _ops()
    N, H, W = 2, 12, 20
    Cs = (8, 5, 16)
    xs = [rnd(N, C, H, W, seed=100 + i).requires_grad_() for i, C in enumerate(Cs)]
    ws = [(rnd(C, seed=110 + i).abs() + 0.5).requires_grad_() for i, C in enumerate(Cs)]
    bs = [rnd(C, seed=120 + i).requires_grad_() for i, C in enumerate(Cs)]
    for mode in (1, 2):
        rms = [rnd(C, seed=130 + i) * 0.1 for i, C in enumerate(Cs)]
        rvs = [rnd(C, seed=140 + i).abs() + 0.5 for i, C in enumerate(Cs)]
        rms_r, rvs_r = [t.clone() for t in rms], [t.clone() for t in rvs]
        ref = torch.cat([F.batch_norm(x, rm, rv, w, b, mode == 1, 0.1, 1e-5) for x, w, b, rm, rv in zip(xs, ws, bs, rms_r, rvs_r)], 1)
        go = rnd(*ref.shape, seed=150)
        gr = torch.autograd.grad(ref, xs + ws + bs, go)
        dev = lambda t: t.detach().to(DEV).requires_grad_(t.requires_grad)
        gx, gw, gb = [dev(t) for t in xs], [dev(t) for t in ws], [dev(t) for t in bs]
        grm, grv = [t.to(DEV) for t in rms], [t.to(DEV) for t in rvs]
        out = ops.norm_act_concat([(x, w, b, rm, rv, None) for x, w, b, rm, rv in zip(gx, gw, gb, grm, grv)], mode)
        close(out, ref, 1e-5, 'concat fwd mode %d' % mode)
        gg = torch.autograd.grad(out, gx + gw + gb, go.to(DEV))
        for a, b_, nm in zip(gg, gr, ['dx'] * 3 + ['dw'] * 3 + ['db'] * 3):
            close(a, b_, 1e-4, 'concat %s mode %d' % (nm, mode))
        for a, b_ in zip(grm + grv, rms_r + rvs_r):
            close(a, b_, 1e-5, 'running stats')


@pytest.mark.parametrize('training', [True, False])
def test_conv_bn_concat_node(training):
    """cat([batch_norm(conv2d(x, w_i, dilation d_i))]) as one autograd node (no cat copies, data gradients summed in the
    transposed-conv epilogue) against torch autograd."""
    ops = _ops()
    N, C, H, W = 2, 8, 20, 36
    dils = (1, 3, 5)
    x = rnd(N, C, H, W, seed=200).requires_grad_()
    ws = [(rnd(C, C, 3, 3, seed=201 + i) * 0.2).requires_grad_() for i in range(3)]
    gs = [(rnd(C, seed=210 + i).abs() + 0.5).requires_grad_() for i in range(3)]
    bs = [rnd(C, seed=220 + i).requires_grad_() for i in range(3)]
    rms = [rnd(C, seed=230 + i) * 0.1 for i in range(3)]
    rvs = [rnd(C, seed=240 + i).abs() + 0.5 for i in range(3)]
    rms_r, rvs_r = [t.clone() for t in rms], [t.clone() for t in rvs]
    ref = torch.cat([F.batch_norm(F.conv2d(x, w, None, 1, d, d), rm, rv, g, b, training, 0.1, 1e-5)
                     for w, g, b, rm, rv, d in zip(ws, gs, bs, rms_r, rvs_r, dils)], 1)
    go = rnd(*ref.shape, seed=250)
    gr = torch.autograd.grad(ref, [x] + ws + gs + bs, go)
    dev = lambda t: t.detach().to(DEV).requires_grad_(t.requires_grad)
    xg, wg, gg_, bg = dev(x), [dev(t) for t in ws], [dev(t) for t in gs], [dev(t) for t in bs]
    rmg, rvg = [t.to(DEV) for t in rms], [t.to(DEV) for t in rvs]
    out = ops.conv_bn_concat(xg, [(w, g, b, rm, rv) for w, g, b, rm, rv in zip(wg, gg_, bg, rmg, rvg)], dils, training)
    close(out, ref, 1e-4, 'conv-bn-cat fwd')
    got = torch.autograd.grad(out, [xg] + wg + gg_ + bg, go.to(DEV))
    for a, b_, nm in zip(got, gr, ['dx'] + ['dw'] * 3 + ['dgamma'] * 3 + ['dbeta'] * 3):
        close(a, b_, 2e-4, 'conv-bn-cat %s' % nm)
    for a, b_ in zip(rmg + rvg, rms_r + rvs_r):
        close(a, b_, 1e-5, 'running stats')
