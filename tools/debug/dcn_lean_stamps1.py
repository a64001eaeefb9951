"""Phase stamps of the all-in-one lean forward kernel (library built with -DDPF_STAMPS)."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops, _lib
import numpy as np
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
dev = 'cuda'
x = torch.randn(4, C, 4, 256, 384, device=dev)
off = torch.randn(4, 81, 4, 256, 384, device=dev) * (0.75 if C == 64 else 1.3)
w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(64, device=dev)
for _ in range(2):
    ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (16 * 128 * 4))()
assert _lib.lib().cdll.dpf_debug_lean_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(16, 128, 4).astype(np.int64)
print('tap: [mfma1+reads | accum1 | mfma2+reads | accum2+slow+table+swaps -> next tap]  (wave 0 / wave 3)')
for t in range(1, 26):
    for wv in (8, 11):
        r = a[wv]
        print('%2d w%d  %5d %5d %5d %5d  total %5d' % (t, wv - 8, r[t, 1] - r[t, 0], r[t, 2] - r[t, 1], r[t, 3] - r[t, 2], r[t + 1, 0] - r[t, 3], r[t + 1, 0] - r[t, 0]), end='   ')
    print()
