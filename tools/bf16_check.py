"""How far the bf16 mixed-precision mode moves the outputs of the golden fixture and of a larger random batch."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
def build(prec):
    opt = load_option(); opt.precision = prec
    m = STEREODPNET(opt); fill_by_recipe(m); return m.cuda().train()
for name, batch in (('golden 2x32x48', None), ('synthetic 2x128x192', synthetic_batch(2, 128, 192, seed=5))):
    if batch is None:
        g = np.load('tests/golden/e2e_train_32x48_b2.npz')
        batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('in_')}
    batch = {k: v.cuda() for k, v in batch.items()}
    r32 = build(32).forward(batch); r16 = build('bf16').forward(batch)
    d = (r32['pred_depth'] - r16['pred_depth']).abs()
    dn = (r32['pred_normal'] - r16['pred_normal']).abs()
    print(name, 'pred_depth max %.3e mean %.3e | pred_normal max %.3e mean %.3e | loss %.5f vs %.5f' % (d.max(), d.mean(), dn.max(), dn.mean(), float(r32['final_loss']), float(r16['final_loss'])))
