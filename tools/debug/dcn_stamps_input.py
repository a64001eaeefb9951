import sys, ctypes, torch, glob
sys.path.insert(0, '.')
from dualpixelface_amd import ops
import numpy as np
C = 64
torch.manual_seed(0)
dev = 'cuda'
x = torch.randn(4, C, 4, 256, 384, device=dev)
off = torch.randn(4, 81, 4, 256, 384, device=dev) * 0.75
w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(64, device=dev)
go = torch.randn(4, 64, 4, 256, 384, device=dev)
import os
os.environ['DPF_DCN_OFF_RS'] = '0'     # stamps of the pk kernel only (the fused offset kernel has none)
for _ in range(2):
    ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
torch.cuda.synchronize()
dll = ctypes.CDLL(os.environ.get('DPF_LIB_PATH') or glob.glob('dualpixelface_amd/libdpf_hip.so')[0])
buf = (ctypes.c_ulonglong * (16 * 128 * 2))()
assert dll.dpf_debug_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(16, 128, 2).astype(np.int64)
print('tap: per wave  len | A: first half (tables or wait) | scatter+mfma | second half')
for t in range(2, 20):
    row = []
    for wv in (0, 7, 8, 15):
        t0, t1 = a[wv, 2 * t]; t2, t3 = a[wv, 2 * t + 1]
        nxt = a[wv, 2 * t + 2, 0]
        row.append('w%d len %6d tab %5d waitB %5d scat %5d waitA %5d' % (wv, nxt - t0, t1 - t0, t2 - t1, t3 - t2, nxt - t3))
    print('%2d ' % t + ' | '.join(row))
