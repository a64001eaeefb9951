"""Run only the weight gradient of one conv_shape_bench shape a few times (for PMC passes): python tools/wgrad_only.py <shape> [reps]"""
import sys, torch
sys.path.insert(0, '.')
sys.argv = [sys.argv[0]] + sys.argv[1:]
from dualpixelface_amd import ops
import importlib.util
src = open('tools/conv_shape_bench.py').read()
shapes = eval(src[src.index('SHAPES = {') + len('SHAPES = '):src.index('\n}\n') + 2])
nm = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, C, D, H, W, K, ks, st, pd, dl = shapes[nm]
x = torch.randn(N, C, D, H, W, device='cuda')
w = torch.randn(K, C, *ks, device='cuda') * 0.1
y = ops.ConvFn.apply(x, w, None, st, pd, dl)
go = torch.randn_like(y)
for _ in range(reps):
    ops._conv_wgrad_raw(go, x, w.shape, st, pd, dl)
torch.cuda.synchronize()
