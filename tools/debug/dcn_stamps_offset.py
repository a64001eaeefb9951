import sys, ctypes, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
from dualpixelface_amd._lib import lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sig = 0.75 if C == 64 else 1.3
torch.manual_seed(0)
dev = 'cuda'
x = torch.randn(4, C, 4, 256, 384, device=dev)
off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig
w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(64, device=dev)
go = torch.randn(4, 64, 4, 256, 384, device=dev)
for _ in range(2):
    ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
torch.cuda.synchronize()
L = lib()
buf = (ctypes.c_ulonglong * (16 * 128 * 2))()
h = L._dll if hasattr(L, '_dll') else None
import glob
import os
dll = ctypes.CDLL(os.environ.get('DPF_LIB_PATH') or glob.glob('dualpixelface_amd/libdpf_hip.so')[0])
assert dll.dpf_debug_stamps(buf) == 0
import numpy as np
a = np.array(buf, dtype=np.uint64).reshape(16, 128, 2).astype(np.int64)
NS = 108 if C == 64 else 81
t0 = a[:, :NS, 0].min()
names = ['smp'] * 8 + ['gcol'] * 4 + ['wgr'] * 4
print('step: per role  start(rel to prev step start)  busy   ; wave 0 (sampler h0), wave 4 (sampler h1), wave 8 (gcol), wave 12 (wgrad)')
for st in range(2, 40):
    row = []
    for wv in (0, 4, 8, 12):
        row.append('%s s%+6d b%5d' % (names[wv], a[wv, st, 0] - a[0, st, 0], a[wv, st, 1] - a[wv, st, 0]))
    print('%3d  step_len %6d | ' % (st, a[0, st + 1, 0] - a[0, st, 0]) + ' | '.join(row))
busy = a[:, 2:NS - 1, 1] - a[:, 2:NS - 1, 0]
print('mean busy per wave:', busy.mean(1).astype(int))
print('mean step length:', (a[0, 3:NS - 1, 0] - a[0, 2:NS - 2, 0]).mean())
