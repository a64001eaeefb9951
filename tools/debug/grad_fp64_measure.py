"""Distance of this implementation's gradients and of the reference's fp32 gradients (fixture) from the fp64 oracle, for the 10 fixture
tensors: python tools/debug/grad_fp64_measure.py [tag]   (run under the kernel-path switches; tests/test_gpu_e2e.py::
test_gradients_no_worse_than_the_reference_vs_fp64 bounds the first by the second)"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_e2e import build_model, load_batch
from oracle import recipe_state
from oracle.stereodpnet import StereoDPNetOracle
tag = sys.argv[1] if len(sys.argv) > 1 else 'train_32x48_b2'
g = np.load('tests/golden/e2e_%s.npz' % tag)
st = recipe_state(dtype=torch.float64)
StereoDPNetOracle(st, training=True).forward({k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith('in_')})['final_loss'].backward()
model = build_model(True)
model.train_step(load_batch(g))
pd = dict(model.named_parameters())
rows = []
for k in g.files:
    if not k.startswith('grad::'):
        continue
    exact = st[k[6:]].grad
    if exact is None or exact.norm().item() < 1e-6:
        continue
    ref32 = torch.from_numpy(g[k]).double()
    mine = pd[k[6:]].grad.detach().cpu().double()
    rows.append((k[6:], ((mine - exact).norm() / exact.norm()).item(), ((ref32 - exact).norm() / exact.norm()).item()))
print(tag, ' '.join('%s mine %.2g ref %.2g |' % (n[-28:], a, b) for n, a, b in rows))
import math
print(tag, 'geometric mean of mine/ref: %.2f' % math.exp(sum(math.log(a / b) for _, a, b in rows) / len(rows)))
