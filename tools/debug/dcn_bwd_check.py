"""Parity of the deformable-conv backward (all four gradients) against the oracle on shapes that take the lean path + timing."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
from oracle import dcn3d
dev = 'cuda'
what = sys.argv[1:] or ['parity', 'time']

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale

if 'parity' in what:
    for cfg in [(2, 35, 64, 4, 6, 12, 1.5), (1, 64, 64, 4, 8, 12, 1.5), (1, 20, 40, 4, 9, 72, 3.0), (1, 16, 24, 3, 7, 44, 4.0), (1, 12, 8, 1, 5, 8, 1.0),
                (1, 36, 33, 2, 37, 20, 6.0), (2, 7, 64, 4, 3, 36, 0.0), (1, 8, 16, 4, 6, 16, -1.0)]:
        B, C, K, D, H, W, sig = cfg
        x = rnd(B, C, D, H, W, seed=70); wt = rnd(K, C, 3, 3, 3, seed=72, scale=0.1); bs = rnd(K, seed=73)
        if sig < 0:      # integer offsets: samples exactly on -1 / the borders (the validity rule cuh:248)
            off = torch.randint(-2, 3, (B, 81, D, H, W), generator=torch.Generator().manual_seed(71)).float()
        else:
            off = rnd(B, 81, D, H, W, seed=71, scale=sig)
        ref = dcn3d.deform_conv3d_forward(x, off, wt, bs)
        go = rnd(*ref.shape, seed=74)
        gr = dcn3d.deform_conv3d_backward(x, off, wt, bs, go)
        g = ops.deform_conv_backward_raw(x.to(dev), wt.to(dev), bs.to(dev), off.to(dev), go.to(dev), (1, 1, 1), (1, 1, 1), (1, 1, 1))
        msg = []
        for a, r, nm in zip(g, gr, ('gi', 'goff', 'gw', 'gb')):
            err = (a.cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-9)
            msg.append('%s %.1e%s' % (nm, err, '' if err < 2e-4 else ' FAIL'))
        print('cfg', cfg, ' '.join(msg))
if 'time' in what:
    for C, sig in ((35, 1.3), (64, 0.75)):
        torch.manual_seed(0)
        x = torch.randn(4, C, 4, 256, 384, device=dev)
        off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig
        w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(64, device=dev)
        go = torch.randn(4, 64, 4, 256, 384, device=dev)
        tb = []
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g = ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
            torch.cuda.synchronize(); tb.append((time.perf_counter() - t0) * 1e3)
        print('C=%d sigma=%.2f  bwd(all) min %.2f med %.2f ms' % (C, sig, min(tb[1:]), sorted(tb[1:])[2]))
