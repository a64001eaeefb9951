"""Forward parity of the lean deformable-conv kernels against the oracle on shapes that take that path + timing at the model's shapes.
argv: [parity] [time]"""
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
                (1, 36, 33, 2, 37, 20, 6.0), (2, 7, 64, 4, 3, 36, 0.0)]:
        B, C, K, D, H, W, sig = cfg
        x = rnd(B, C, D, H, W, seed=70); off = rnd(B, 81, D, H, W, seed=71, scale=sig); wt = rnd(K, C, 3, 3, 3, seed=72, scale=0.1); bs = rnd(K, seed=73)
        ref = dcn3d.deform_conv3d_forward(x, off, wt, bs)
        y = ops.deform_conv_forward_raw(x.to(dev), wt.to(dev), bs.to(dev), off.to(dev), (1, 1, 1), (1, 1, 1), (1, 1, 1)).cpu()
        err = (y - ref).abs().max().item() / ref.abs().max().item()
        print('cfg', cfg, 'rel err %.2e' % err, 'OK' if err < 1e-4 else 'FAIL')
if 'time' in what:
    for C, sig in ((35, 1.3), (64, 0.75)):
        torch.manual_seed(0)
        x = torch.randn(4, C, 4, 256, 384, device=dev)
        off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig
        w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(64, device=dev)
        tf = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            tf.append((t1 - t0) * 1e3)
        print('C=%d sigma=%.2f  fwd min %.2f med %.2f ms' % (C, sig, min(tf[1:]), sorted(tf[1:])[2]))
