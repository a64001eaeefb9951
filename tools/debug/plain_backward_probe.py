"""Plain loss.backward() with the two feature passes on two streams: preset .grad arena views vs .grad = None."""
import os, sys, torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
import dualpixelface_amd.stereodpnet as sd
batch = {k: v.cuda() for k, v in synthetic_batch(2, 32, 48, seed=11).items()}
def run(two, preset):
    sd.FEATURES_TWO_STREAMS = two
    m = STEREODPNET(load_option()); fill_by_recipe(m); m = m.cuda().train()
    m._two_streams_ok = two
    if preset:
        m.flat_gradients(zero=True)
    else:
        for p in m.parameters():
            p.grad = None
    m.forward(batch)['final_loss'].backward()
    torch.cuda.synchronize()
    name = 'feature_extraction.firstconv.0.0.weight'
    return dict(m.named_parameters())[name].grad.detach().clone()
base = run(False, True)
for two in (False, True):
    for preset in (True, False):
        for rep in range(3):
            g = run(two, preset)
            print('two %5s preset %5s rep %d: firstconv.0.0.weight max diff %.3e (scale %.3e)' % (two, preset, rep, (g - base).abs().max().item(), base.abs().max().item()))
