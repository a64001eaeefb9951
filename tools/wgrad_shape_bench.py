"""Time the weight-gradient kernel alone on a few shapes."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
SHAPES = {'hg32': (4, 32, 8, 256, 384, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
          'fe32': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1)),
          'off81': (4, 64, 4, 256, 384, 81, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
          'hg_s2': (4, 32, 8, 256, 384, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
          'n_d8': (16, 64, 1, 256, 384, 64, (1, 3, 3), (1, 1, 1), (0, 8, 8), (1, 8, 8)),
          'n_d4': (16, 96, 1, 256, 384, 64, (1, 3, 3), (1, 1, 1), (0, 4, 4), (1, 4, 4)),
          'n_d2': (16, 96, 1, 256, 384, 96, (1, 3, 3), (1, 1, 1), (0, 2, 2), (1, 2, 2)),
          'fe_d5': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 5, 5), (1, 5, 5)),
          'fe_d3': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 3, 3), (1, 3, 3)),
          'fe2_d5': (4, 32, 1, 256, 384, 32, (1, 3, 3), (1, 1, 1), (0, 5, 5), (1, 5, 5)),
          'fe3_d3': (4, 64, 1, 128, 192, 64, (1, 3, 3), (1, 1, 1), (0, 3, 3), (1, 3, 3))}
for nm in [a for a in sys.argv[1:] if not a.startswith('--')] or list(SHAPES):
    N, C, D, H, W, K, ks, st, pd, dl = SHAPES[nm]
    x = torch.randn(N, C, D, H, W, device='cuda')
    od = [(i + 2 * p - (d * (k - 1) + 1)) // s + 1 for i, k, s, p, d in zip((D, H, W), ks, st, pd, dl)]
    g = torch.randn(N, K, *od, device='cuda')
    ts = []
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops._conv_wgrad_raw(g, x, (K, C) + ks, st, pd, dl)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts[1:]); fl = 2.0 * g.numel() * C * ks[0] * ks[1] * ks[2]
    err = ''
    if '--check' in sys.argv or True:
        ref = torch.nn.grad.conv3d_weight(x[:1].double().cpu(), (K, C) + ks, g[:1].double().cpu(), st, pd, dl) if N * D * H * W <= 4 * 512 * 768 else None
        if ref is not None:
            got = ops._conv_wgrad_raw(g[:1].contiguous(), x[:1].contiguous(), (K, C) + ks, st, pd, dl).double().cpu()
            err = ' rel err vs fp64 (1 sample) %.2e' % ((got - ref).abs().max() / ref.abs().max()).item()
    print('%s wgrad %.3f ms %.1f TF%s' % (nm, t * 1e3, fl / t * 1e-12, err), flush=True)
