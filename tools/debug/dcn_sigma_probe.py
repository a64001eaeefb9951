"""How much of the deformable-conv kernels' time is the scattered LDS gather / scatter: the same launches with offsets of different spread
(sigma = 0: every wave reads / adds consecutive cells; the model's layers have sigma ~ 1.3 and ~ 0.75)."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
dev = 'cuda'
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
x = torch.randn(4, C, 4, 256, 384, device=dev)
w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(64, device=dev)
go = torch.randn(4, 64, 4, 256, 384, device=dev)
base = torch.randn(4, 81, 4, 256, 384, device=dev)
smooth = torch.nn.functional.avg_pool3d(base, (1, 9, 9), 1, (0, 4, 4)) * 9.0     # spatially correlated field with unit-ish spread
for name, off in (('sigma 0', base * 0), ('sigma 0.3', base * 0.3), ('sigma 0.75', base * 0.75), ('sigma 1.3', base * 1.3),
                  ('smooth field, sigma %.2f' % float(smooth.std()), smooth), ('constant 0.4', base * 0 + 0.4)):
    off = off.contiguous()
    tf, tb = [], []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tf.append((t1 - t0) * 1e3); tb.append((t2 - t1) * 1e3)
    print('C=%d %-28s fwd %.2f ms   bwd %.2f ms' % (C, name, min(tf[1:]), min(tb[1:])))
