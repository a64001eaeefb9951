"""Loss over N train steps on one fixed synthetic batch (the plugin's fused step, Adam, lr from the shipped config):
python tools/debug/loss_curve.py [steps] [H] [W] [batch]   -- run under DPF_F32_X9=0 / default to compare the fp32 matrix paths."""
import sys, torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W = int(sys.argv[3]) if len(sys.argv) > 3 else 384
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
torch.manual_seed(0)
m = STEREODPNET(load_option()); fill_by_recipe(m); m.cuda().train()
batch = {k: v.cuda() for k, v in synthetic_batch(B, H, W, seed=3).items()}
out = []
for i in range(steps):
    res = m.train_step(batch)
    if i % 5 == 0 or i == steps - 1:
        out.append('%d:%.5f' % (i, float(res['final_loss'])))
print(' '.join(out))
