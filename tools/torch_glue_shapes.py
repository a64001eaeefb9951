"""Which torch kernels (autograd's gradient sums, copies, fills) run in one eager train step, by tensor shape.
python tools/torch_glue_shapes.py [B H W]  ->  table of aten ops with device time, sorted; one line per (op, input shapes)."""
import os, sys
sys.path.insert(0, '.')
import torch
from torch.profiler import profile, ProfilerActivity
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import synthetic_batch
B, H, W = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (4, 1024, 1536)
os.environ['DPF_STEP_GRAPH'] = '0'
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = STEREODPNET(load_option()).to(dev)
batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, W, seed=0).items()}
for _ in range(2):
    model.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    model.train_step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = getattr(e, 'self_cuda_time_total', 0)
    if t > 0 and e.key.startswith('aten::'):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('aten ops with device time in one step: %.3f ms, %d calls' % (tot / 1e3, sum(r[1] for r in rows)))
for t, n, k, s in rows[:70]:
    print('%9.1f us %4d x  %-28s %s' % (t, n, k, s))
