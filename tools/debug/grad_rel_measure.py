"""Distance of the 10 fixture gradients (tests/golden/e2e_*.npz, produced by the imported reference in fp32) from this implementation's,
per fixture -- the numbers behind GRAD_REL_MEASURED in tests/test_gpu_e2e.py.  Run under the kernel-path switches to see how far
legitimate changes of the fp32 summation order move them:  python tools/debug/grad_rel_measure.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_e2e import build_model, load_batch
out = {}
for tag in ('train_32x48_b2', 'train_64x96_b1', 'train_128x128_b2'):
    g = np.load('tests/golden/e2e_%s.npz' % tag)
    model = build_model(True)
    model.train_step(load_batch(g))
    pd = dict(model.named_parameters())
    row = {}
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        ref = torch.from_numpy(g[k]).double()
        if ref.norm().item() < 1e-6:
            continue
        mine = pd[k[6:]].grad.detach().cpu().double()
        row[k[6:]] = float('%.2g' % ((mine - ref).norm() / ref.norm()).item())
    out[tag] = row
    print(tag, row, flush=True)
