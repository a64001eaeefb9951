"""Time (and check against torch's own GPU convolution) one convolution shape through the C-ABI: forward, data gradient,
weight gradient.  usage: python tools/conv_shape_bench.py [--check] [names...]   (env knobs: DPF_IGEMM2, DPF_IGEMM3, DPF_F32_X9, ...)"""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
import torch.nn.functional as F
dev = 'cuda'
SHAPES = {
    'hg32': (4, 32, 8, 256, 384, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'hg_s2': (4, 32, 8, 256, 384, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    'hg64': (4, 64, 4, 128, 192, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'hg64_s2': (4, 64, 4, 128, 192, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    'hg64q': (4, 64, 2, 64, 96, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'cv64_32': (4, 64, 8, 256, 384, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'fe32': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'fe32q': (4, 32, 1, 256, 384, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'fe32d5': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 5, 5), (1, 5, 5)),
    'fe96_32': (4, 96, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'fe64': (4, 64, 1, 128, 192, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'fe192_64': (4, 192, 1, 128, 192, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'anm96d2': (16, 96, 1, 256, 384, 96, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
    'anm64d8': (16, 64, 1, 256, 384, 64, (1, 3, 3), (1, 1, 1), (0, 8, 8), (1, 8, 8)),
    'off81': (4, 64, 4, 256, 384, 81, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'off81a': (4, 35, 4, 256, 384, 81, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'pw32': (4, 32, 1, 256, 384, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    'pw32x3': (4, 32, 3, 256, 384, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    'pw64': (4, 64, 1, 128, 192, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    'pw128': (4, 128, 1, 64, 96, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    'pw128_32': (4, 128, 1, 64, 96, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    'pw32s2': (4, 32, 1, 512, 768, 32, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),
    'pw64_128s2': (4, 64, 1, 128, 192, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 1, 1)),
}
args = sys.argv[1:]
check = '--check' in args
# --noguard: the f16-component path without its range guards (round 5's kernels); --relu: half of the inputs exactly zero (what a
# post-ReLU tensor looks like: the guard's position test then meets positions with a single small non-zero channel)
if '--noguard' in args:
    from dualpixelface_amd._lib import lib
    lib().call('dpf_debug_set_range_guard', 0)
relu = '--relu' in args
names = [a for a in args if not a.startswith('--')] or list(SHAPES)
for nm in names:
    N, C, D, H, W, K, ks, st, pd, dl = SHAPES[nm]
    torch.manual_seed(0)
    x = torch.randn(N, C, D, H, W, device=dev)
    x = (x.relu() if relu else x).requires_grad_()
    w = (torch.randn(K, C, *ks, device=dev) * 0.1).requires_grad_()
    y = ops.ConvFn.apply(x, w, None, st, pd, dl)
    go = torch.randn_like(y)
    flops = 2.0 * y.numel() * C * ks[0] * ks[1] * ks[2]
    res = []
    if check:
        xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
        yr = F.conv3d(xr, wr, None, st, pd, dl)
        gxr, gwr = torch.autograd.grad(yr, (xr, wr), go)
        gx, gw = torch.autograd.grad(y, (x, w), go, retain_graph=True)
        rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()
        res.append('err fwd %.1e dgrad %.1e wgrad %.1e' % (rel(y, yr), rel(gx, gxr), rel(gw, gwr)))
        del xr, wr, yr, gxr, gwr, gx, gw
    for what in ('fwd', 'dgrad', 'wgrad'):
        ts = []
        REPS = 10                                   # back-to-back launches per timing: amortises the ~40 us launch + sync floor
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(REPS):
                if what == 'fwd':
                    ops.ConvFn.apply(x.detach(), w.detach(), None, st, pd, dl)
                elif what == 'dgrad':
                    ops._conv_transpose_raw(go, w.detach(), None, x.shape[2:], w.shape[2:], st, pd, dl)
                else:
                    ops._conv_wgrad_raw(go, x.detach(), w.shape, st, pd, dl)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / REPS)
        t = min(ts[1:])
        res.append('%s %.3f ms %.1f TF' % (what, t * 1e3, flops / t * 1e-12))
    print('%-9s' % nm, ' | '.join(res), flush=True)
    del x, w, y, go
