import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dualpixelface_amd import load_option, ops
import dualpixelface_amd.stereodpnet as sdn
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import synthetic_batch
from dualpixelface_amd._lib import lib
B, H, W = [int(a) for a in sys.argv[1:4]]
ops.WGRAD_ASYNC = False; sdn.FEATURES_TWO_STREAMS = False
os.environ['DPF_STEP_GRAPH'] = '0'
dev = torch.device('cuda', 0)
batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, W, seed=0).items()}
torch.manual_seed(3)
base = STEREODPNET(load_option()).to(dev)
sd = {k: v.clone() for k, v in base.state_dict().items()}
out = {}
for path in (2, 0):
    lib().call('dpf_set_f32_matrix_path', path)
    model = STEREODPNET(load_option()).to(dev)
    model.load_state_dict(sd, strict=True)
    res = model.train_step(batch)
    torch.cuda.synchronize()
    g = model.flat_gradients(zero=False).clone()
    out[path] = (g, res['pred_depth'].detach().clone(), model._layout)
    bad = [(name, int((~torch.isfinite(g[off:off + numel])).sum())) for name, off, numel, _ in model._layout if not torch.isfinite(g[off:off + numel]).all()]
    print('path', path, 'loss', float(res['final_loss']), 'non-finite gradient tensors:', len(bad), bad[:12])
g2, d2, layout = out[2]; g0, d0, _ = out[0]
print('disparity max diff', (d2 - d0).abs().max().item(), 'at', (d2 - d0).abs().argmax().item(), 'shape', tuple(d2.shape))
dd = (d2 - d0).abs()
print('disparity diff > 2e-3:', int((dd > 2e-3).sum()), 'of', dd.numel(), ' per head max', dd.amax(dim=(0, 2, 3)).tolist())
rows = dd.amax(dim=(0, 1, 3)); cols = dd.amax(dim=(0, 1, 2))
print('rows with diff > 1e-2:', (rows > 1e-2).nonzero().flatten().tolist()[:40])
print('cols with diff > 1e-2:', (cols > 1e-2).nonzero().flatten().tolist()[:40])
worst = []
for name, off, numel, _ in layout:
    a, b = g0[off:off + numel], g2[off:off + numel]
    if torch.isfinite(b).all() and a.norm() > 0:
        worst.append((((a - b).norm() / a.norm()).item(), name))
worst.sort(reverse=True)
print('worst finite parameters', worst[:8])
