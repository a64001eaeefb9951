"""Does plugin.train_step's HIP-graph capture survive on this runtime?  argv: wgrad_async(0/1) two_streams(0/1) [size]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from dualpixelface_amd import ops
import dualpixelface_amd.stereodpnet as sdn
from test_gpu_e2e import build_model
from dualpixelface_amd.recipe import synthetic_batch
ops.WGRAD_ASYNC = sys.argv[1] == '1'
sdn.FEATURES_TWO_STREAMS = sys.argv[2] == '1'
H, W, B = (int(v) for v in (sys.argv[3].split('x') if len(sys.argv) > 3 else ('128', '128', '2')))
batch = {k: v.cuda() for k, v in synthetic_batch(B, H, W, seed=0).items()}
model = build_model(True)
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = model.train_step(batch)
    torch.cuda.synchronize()
    print('step', i, 'loss %.6f' % float(res['final_loss']), '%.1f ms' % ((time.perf_counter() - t0) * 1e3), 'graph' if (getattr(model, '_graph_state', None) or {}).get('graph') is not None else 'eager', flush=True)
