import torch, sys
sys.path.insert(0,'.')
from dualpixelface_amd import load_option, ops
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import synthetic_batch
torch.manual_seed(1)
m=STEREODPNET(load_option()).to('cuda'); m.train()
orig=ops.deform_conv3d
def spy(x, off, w, b, *a, **k):
    o=off.detach().abs()
    print('offset |.| mean %.3f  p50 %.3f p90 %.3f p99 %.3f max %.2f ; frac>2: %.3f frac>3: %.3f frac>4 %.3f'%(o.mean(), o.flatten()[::97].quantile(0.5), o.flatten()[::97].quantile(0.9), o.flatten()[::97].quantile(0.99), o.max(), (o>2).float().mean(), (o>3).float().mean(), (o>4).float().mean()))
    return orig(x, off, w, b, *a, **k)
ops.deform_conv3d=spy
b={k:v.to('cuda') for k,v in synthetic_batch(1,512,768).items()}
for i in range(2): m.train_step(b)
