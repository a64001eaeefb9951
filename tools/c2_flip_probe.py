"""How often, and how far, does the c2-shape normal head differ from the CPU oracle because a quarter-resolution pixel picks a
neighbouring cost level (top-k at a level boundary)?  Runs the oracle once and the HIP model N times; prints per run the number
of flipped pixels, the largest normal error outside / inside their receptive fields and the loss deviations."""
import sys, torch
sys.path.insert(0, '.')
from oracle import recipe_state
from oracle.stereodpnet import StereoDPNetOracle
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = synthetic_batch(1, 512, 768, seed=21, mask_mode='bern')
orc = StereoDPNetOracle(recipe_state(requires_grad=False), training=True)
with torch.no_grad():
    ref = orc.forward(batch)
idx_cpu = orc.taps['anm_idx'].long()
gb = {k: v.to('cuda') for k, v in batch.items()}
for run in range(N):
    model = STEREODPNET(load_option())
    fill_by_recipe(model)
    model.to('cuda').train()
    res = model.train_step(gb)
    flipped = (model.last_anm_idx.cpu().long() != idx_cpu).any(1, keepdim=True).float()
    err = (res['pred_normal'].detach().cpu().double() - ref['pred_normal'].double()).abs()
    out = []
    for R in (16, 24, 32, 48):
        near = torch.nn.functional.max_pool2d(flipped, 2 * R + 1, 1, R)
        near = torch.nn.functional.interpolate(near, scale_factor=4, mode='nearest').bool()
        out.append('R%d %.1e' % (R, float(err.masked_fill(near.unsqueeze(2), 0.0).max())))
    dd = float((res['pred_depth'].detach().cpu() - ref['pred_depth']).abs().max())
    print('run %2d flipped %d  err max %.2e  outside: %s  | depth %.1e  dcos %.1e dfinal %.1e' % (
        run, int(flipped.sum()), float(err.max()), ' '.join(out), dd,
        abs(float(res['cosine_loss']) - float(ref['cosine_loss'])) / float(ref['cosine_loss']),
        abs(float(res['final_loss']) - float(ref['final_loss'])) / float(ref['final_loss'])), flush=True)
