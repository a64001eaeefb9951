import os, sys, torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
import dualpixelface_amd.stereodpnet as sd
batch = {k: v.cuda() for k, v in synthetic_batch(2, 32, 48, seed=11).items()}
out = {}
for two in (False, True, True):
    sd.FEATURES_TWO_STREAMS = two
    m = STEREODPNET(load_option()); fill_by_recipe(m); m = m.cuda().train()
    m.flat_gradients(zero=True)
    res = m.forward(batch)
    res['final_loss'].backward()
    torch.cuda.synchronize()
    g = m.flat_gradients(zero=False).detach().clone()
    m2 = STEREODPNET(load_option()); fill_by_recipe(m2); m2 = m2.cuda().train()
    m2.train_step(batch)
    torch.cuda.synchronize()
    g2 = m2.flat_gradients(zero=False).detach().clone()
    print('two_streams', two, 'plain-backward vs train_step: max diff %.3e (scale %.3e)' % ((g - g2).abs().max().item(), g2.abs().max().item()))
    out[two] = (g, g2)
print('plain: two vs one %.3e ; train_step: two vs one %.3e' % ((out[True][0] - out[False][0]).abs().max().item(), (out[True][1] - out[False][1]).abs().max().item()))
