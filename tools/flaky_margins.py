"""Measured margins of the tolerance-based GPU tests that depend on atomics ordering (run several times)."""
import sys, tempfile, torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe
from dualpixelface_amd.synthetic_data import synthetic_loader
from dualpixelface_amd.trainer import Trainer
def model(opt):
    m = STEREODPNET(opt); fill_by_recipe(m); return m.to('cuda').train()
for rep in range(6):
    opt = load_option(); opt.epoch, opt.init_lr, opt.scheduler = 2, 1e-3, 'explr'
    loader = synthetic_loader(4, 32, 48, batch_size=2, seed=3)
    d = tempfile.mkdtemp()
    a = model(opt); ta = Trainer(opt, d + '/a', rank=0, world_size=1); ta.fit(a, loader, None)
    opt.load_model = ta.checkpoint_path(0)
    b = model(opt); tb = Trainer(opt, d + '/b', rank=0, world_size=1); tb.fit(b, loader, None)
    diff = (a.flat_parameters().cpu() - b.flat_parameters().cpu()).abs()
    print('resume: frac>1e-5 %.2e  max %.3e' % ((diff > 1e-5).float().mean().item(), diff.max().item()), flush=True)
