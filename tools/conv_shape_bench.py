"""Time one convolution shape (forward / data gradient / weight gradient) through the C-ABI."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
import torch.nn.functional as F
dev = 'cuda'
SHAPES = {
    'hg32': (4, 32, 8, 256, 384, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    'hg_s2': (4, 32, 8, 256, 384, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    'fe32': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'fe32q': (4, 32, 1, 256, 384, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'fe64': (4, 64, 1, 128, 192, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
    'off81': (4, 64, 4, 256, 384, 81, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
}
names = sys.argv[1:] or list(SHAPES)
for nm in names:
    N, C, D, H, W, K, ks, st, pd, dl = SHAPES[nm]
    x = torch.randn(N, C, D, H, W, device=dev).requires_grad_()
    w = (torch.randn(K, C, *ks, device=dev) * 0.1).requires_grad_()
    y = ops.ConvFn.apply(x, w, None, st, pd, dl)
    go = torch.randn_like(y)
    flops = 2.0 * y.numel() * C * ks[0] * ks[1] * ks[2]
    res = []
    for what in ('fwd', 'dgrad', 'wgrad'):
        ts = []
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if what == 'fwd':
                ops.ConvFn.apply(x.detach(), w.detach(), None, st, pd, dl)
            elif what == 'dgrad':
                torch.autograd.grad(y, x, go, retain_graph=True)
            else:
                torch.autograd.grad(y, w, go, retain_graph=True)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        res.append('%s %.3f ms %.1f TF' % (what, t * 1e3, flops / t * 1e-12))
    print(nm, ' | '.join(res), flush=True)
