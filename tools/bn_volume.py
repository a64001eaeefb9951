"""Bytes streamed by the normalisation kernels in one train step (to price them against the HBM roofline)."""
import sys, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.config import load_option
from dualpixelface_amd.recipe import synthetic_batch
tot = {}
orig = ops.norm_act
def wrapped(x, *a, **k):
    mode = k.get('mode', a[7] if len(a) > 7 else 0)
    has_res = (len(a) > 3 and a[3] is not None) or k.get('res') is not None
    key = (mode, has_res)
    e = tot.setdefault(key, [0, 0])
    e[0] += 1; e[1] += x.numel() * 4
    return orig(x, *a, **k)
ops.norm_act = wrapped
import dualpixelface_amd.stereodpnet as sd
sd.ops.norm_act = wrapped
B, H, W = 4, 1024, 1536
model = STEREODPNET(load_option()).cuda()
batch = {k: v.cuda() for k, v in synthetic_batch(B, H, W, seed=0).items()}
model.train_step(batch, None)
torch.cuda.synchronize()
for k, (n, b) in sorted(tot.items()):
    print('mode %d res %d: calls %d  tensor bytes %.2f GB' % (k[0], k[1], n, b / 1e9))
