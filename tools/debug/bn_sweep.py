"""Sweep of the BatchNorm kernels' chunk sizes: time of forward apply / backward on the big hourglass tensor."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
x = torch.randn(4, 32, 8, 256, 384, device='cuda', requires_grad=True)
w = torch.ones(32, device='cuda', requires_grad=True); b = torch.zeros(32, device='cuda', requires_grad=True)
rm = torch.zeros(32, device='cuda'); rv = torch.ones(32, device='cuda')
go = torch.randn_like(x)
def run():
    y = ops.norm_act(x, w, b, None, None, None, rm, rv, 1, ops.ACT_RELU)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = ops.norm_act(x, w, b, None, None, None, rm, rv, 1, ops.ACT_RELU)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    y.backward(go)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
for _ in range(2): run()
f, bw = zip(*[run() for _ in range(5)])
print('fwd (stats + apply) %.3f ms   bwd (reduce + apply) %.3f ms' % (min(f), min(bw)))
