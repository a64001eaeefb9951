"""Largest relative gradient error against the fp64 oracle over repeated runs (margin of tests/test_gpu_e2e.py's 5e-2)."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import recipe_state
from oracle.stereodpnet import StereoDPNetOracle
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe
g = np.load('tests/golden/e2e_train_32x48_b2.npz')
st = recipe_state(dtype=torch.float64)
b64 = {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith('in_')}
StereoDPNetOracle(st, training=True).forward(b64)['final_loss'].backward()
batch = {k[3:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith('in_')}
for rep in range(8):
    m = STEREODPNET(load_option()); fill_by_recipe(m); m.cuda().train()
    m.train_step(batch)
    pd = dict(m.named_parameters()); rels = []
    for name, off, numel, shape in m._layout:
        ref = st[name].grad
        if ref is None or ref.norm().item() < 1e-6: continue
        rels.append(((pd[name].grad.detach().cpu().double() - ref).norm().item() / ref.norm().item(), numel, name))
    rels.sort(reverse=True)
    print('top:', ['%.3f (n=%d) %s' % (r, n, nm[-40:]) for r, n, nm in rels[:3]], flush=True)
