import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_e2e import build_model, load_batch
ge = np.load('tests/golden/e2e_fixmode_eval_32x48_b2.npz')
g2 = np.load('tests/golden/e2e_eval_32x48_b2.npz')
for name, g, kw in (('fix eval', ge, dict(asm_grid_cache_compat=False)), ('compat eval', g2, {})):
    model = build_model(False, **kw)
    with torch.no_grad():
        res = model(load_batch(g))
    e = (res['pred_depth'].cpu().double() - torch.from_numpy(g['pred_depth']).double()).abs()
    print(name, 'pred_depth max err %.3e mean %.3e' % (e.max().item(), e.mean().item()))
