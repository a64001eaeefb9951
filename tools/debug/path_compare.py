"""One forward + backward of the bench workload under each fp32 matrix path (dpf_set_f32_matrix_path 0 / 1 / 2): losses and the relative L2
distance of every parameter gradient to path 0 (the fp32 MFMA instruction).   python tools/debug/path_compare.py [B H W] [steps]"""
import sys
import torch
sys.path.insert(0, '.')
from dualpixelface_amd import load_option, ops
from dualpixelface_amd._lib import lib
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import synthetic_batch

B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 1024, 1536)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device('cuda:0')
import os
os.environ['DPF_STEP_GRAPH'] = '0'
opt = load_option('train_faceDP')
out = {}
for path in (0, 1, 2):
    lib().call('dpf_set_f32_matrix_path', path)
    torch.manual_seed(0)
    model = STEREODPNET(opt).to(dev)
    model.train()
    losses = []
    for s in range(steps):
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_batch(B, H, W, seed=21 + s).items()}
        res = model.train_step(batch)
        losses.append(float(res['final_loss']))
    torch.cuda.synchronize()
    out[path] = (losses, {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None})
    del model
    torch.cuda.empty_cache()
for path in (0, 1, 2):
    print('path', path, 'losses', ['%.7f' % l for l in out[path][0]])
ref = out[0][1]
for path in (1, 2):
    d = sorted(((out[path][1][n] - ref[n]).norm().item() / max(ref[n].norm().item(), 1e-30), n) for n in ref)
    print('path %d vs 0: median rel L2 %.2e, worst: %s' % (path, d[len(d) // 2][0], [('%.2e' % a, n) for a, n in d[-6:]]))
