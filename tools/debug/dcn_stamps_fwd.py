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
for _ in range(2):
    ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
torch.cuda.synchronize()
dll = ctypes.CDLL(glob.glob('dualpixelface_amd/libdpf_hip.so')[0])
buf = (ctypes.c_ulonglong * (16 * 128 * 2))()
assert dll.dpf_debug_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(16, 128, 2).astype(np.int64)
print('tap: sampler wave 0: step len, busy | MFMA wave 4: start offset (rel sampler start), busy')
for t in range(2, 20):
    print('%2d  smp len %6d busy %6d | mfma start %+6d busy %6d | smp3 busy %6d mfma7 busy %6d' % (t, a[0, t + 1, 0] - a[0, t, 0], a[0, t, 1] - a[0, t, 0],
          a[4, t, 0] - a[0, t, 0], a[4, t, 1] - a[4, t, 0], a[3, t, 1] - a[3, t, 0], a[7, t, 1] - a[7, t, 0]))
