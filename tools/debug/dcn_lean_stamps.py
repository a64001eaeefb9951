"""In-kernel step stamps of the lean deformable-conv forward (library built with -DDPF_STAMPS: DPF_LIB_PATH=.../libdpf_hip_stamps.so)."""
import sys, ctypes, os, torch
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
dll = _lib.lib().cdll
buf = (ctypes.c_ulonglong * (16 * 128 * 4))()
assert dll.dpf_debug_lean_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(16, 128, 4).astype(np.int64)
nw = 8 if a[4:8].any() else 4
ns = nw // 2
print('step: sampler w0 [len | sample | barrier wait]   matrix w%d [start rel | busy]' % ns)
for t in range(1, 27):
    print('%2d  smp len %6d samp %6d wait %6d | mfma start %+6d busy %6d | smp%d samp %6d' % (
        t, a[0, t, 0] - a[0, t - 1, 0], a[0, t, 1] - a[0, t, 0], a[0, t, 2] - a[0, t, 1],
        a[ns, t, 0] - a[0, t, 0], a[ns, t, 1] - a[ns, t, 0], ns - 1, a[ns - 1, t, 1] - a[ns - 1, t, 0]))
