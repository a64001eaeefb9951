import os, sys, torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
from dualpixelface_amd.distributed import make_reducer
import dualpixelface_amd.stereodpnet as sd
batch = {k: v.cuda() for k, v in synthetic_batch(2, 32, 48, seed=11).items()}
def run(two, staged):
    sd.FEATURES_TWO_STREAMS = two
    m = STEREODPNET(load_option()); fill_by_recipe(m); m = m.cuda().train()
    red = None
    if staged:
        red = make_reducer(m)
        red.world_size = 2          # pretend: the hooks run, the collectives are no-ops (_launch only acts when dist is initialised)
        red._launch = lambda bi: None
    m.train_step(batch, red)
    torch.cuda.synchronize()
    if red is not None:
        print('   log', red.log)
    return m.flat_gradients(zero=False).detach().clone()
base = run(False, False)
for two in (False, True):
    for staged in (False, True):
        g = run(two, staged)
        print('two_streams %s staged %s: max diff vs base %.3e (scale %.3e)' % (two, staged, (g - base).abs().max().item(), base.abs().max().item()))
