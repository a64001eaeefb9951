"""Deformable-conv kernel time against the offset spread (how much the far-sample slow path costs)."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
dev = 'cuda'
for C in (35, 64):
    for sig in (0.05, 0.4, 0.75, 1.3, 2.0):
        torch.manual_seed(0)
        x = torch.randn(4, C, 4, 256, 384, device=dev)
        off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig
        w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
        b = torch.zeros(64, device=dev)
        go = torch.randn(4, 64, 4, 256, 384, device=dev)
        tf, tb = [], []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            y = ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
            torch.cuda.synchronize(); t1 = time.perf_counter()
            g = ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
            torch.cuda.synchronize(); t2 = time.perf_counter()
            tf.append((t1 - t0) * 1e3); tb.append((t2 - t1) * 1e3)
        print('C=%d sigma=%.2f  fwd %.2f ms   bwd(all) %.2f ms' % (C, sig, min(tf[1:]), min(tb[1:])), flush=True)
