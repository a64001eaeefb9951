"""How often does the range guard of igemm3_x9_kernel act in the bench step?  (library built with -DDPF_STAMPS:
DPF_LIB_PATH=dualpixelface_amd/libdpf_hip_stamps.so python tools/x9_pass_counts.py [B H W])  Counts over ONE eager train step:
chunk passes, extra passes over deferred positions, accumulator rescales, tiles."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from dualpixelface_amd import load_option, _lib
from dualpixelface_amd.plugin import STEREODPNET
from dualpixelface_amd.recipe import synthetic_batch
B, H, W = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (4, 1024, 1536)
os.environ['DPF_STEP_GRAPH'] = '0'
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = STEREODPNET(load_option()).to(dev)
batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, W, seed=0).items()}
buf = (ctypes.c_ulonglong * 4)()
for step in range(3):
    model.train_step(batch)
    torch.cuda.synchronize()
    assert _lib.lib().cdll.dpf_debug_x9_passes(buf) == 0
    c = list(buf)
    print('step %d: %d tiles, %d chunk passes, %d extra passes over deferred positions (%.3f %%), %d accumulator rescales (%.2f per tile)'
          % (step, c[3], c[0], c[1], 100.0 * c[1] / max(c[0], 1), c[2], c[2] / max(c[3], 1)))
