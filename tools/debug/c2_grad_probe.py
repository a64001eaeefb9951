"""One-off probe behind the budgets of tests/test_gpu_e2e.py::test_c2_shape_forward_loss_and_gradients_vs_cpu_oracle: at 1 x 512 x 768, the
distance of (a) the fp32 CPU oracle and (b) the HIP path to the fp64 CPU oracle, per compared parameter."""
import sys, time, torch
sys.path.insert(0, '.')
from oracle import recipe_state
from oracle.stereodpnet import StereoDPNetOracle
from dualpixelface_amd import load_option
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
torch.set_num_threads(16)
batch = synthetic_batch(1, 512, 768, seed=21, mask_mode='bern')
t0 = time.time()
st64 = recipe_state(dtype=torch.float64)
o64 = StereoDPNetOracle(st64, training=True)
o64.forward({k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()})['final_loss'].backward()
print('fp64 oracle %.0f s' % (time.time() - t0)); t0 = time.time()
st32 = recipe_state()
o32 = StereoDPNetOracle(st32, training=True)
o32.forward(batch)['final_loss'].backward()
print('fp32 oracle %.0f s' % (time.time() - t0))
m = STEREODPNET(load_option()); fill_by_recipe(m); m.to('cuda')
m.train_step({k: v.to('cuda') for k, v in batch.items()})
pd = dict(m.named_parameters())
flip = int((m.last_anm_idx.cpu().long() != o64.taps['anm_idx'].long()).any(1).sum()), int((o32.taps['anm_idx'].long() != o64.taps['anm_idx'].long()).any(1).sum())
print('flipped ANM pixels vs fp64: hip %d, fp32 oracle %d' % flip)
rows = []
for name, p in pd.items():
    e = st64[name].grad if name in st64 else None
    if e is None or p.grad is None or e.norm().item() < 1e-9:
        continue
    a = ((st32[name].grad.double() - e).norm() / e.norm()).item()
    b = ((p.grad.detach().cpu().double() - e).norm() / e.norm()).item()
    c = ((p.grad.detach().cpu().double() - st32[name].grad.double()).norm() / e.norm()).item()
    rows.append((name, a, b, c))
rows.sort(key=lambda r: -r[3])
print('%-60s %10s %10s %10s' % ('parameter', 'cpu32-f64', 'hip-f64', 'hip-cpu32'))
for r in rows[:25]:
    print('%-60s %10.3e %10.3e %10.3e' % r)
import statistics
print('median', statistics.median(r[1] for r in rows), statistics.median(r[2] for r in rows), statistics.median(r[3] for r in rows), 'n', len(rows))
