"""Phase stamps of igemm3_x9_kernel (library built with -DDPF_STAMPS: DPF_LIB_PATH=dualpixelface_amd/libdpf_hip_stamps.so).
usage: python tools/x9_stamps.py <shape name of tools/conv_shape_bench.py>"""
import sys, ctypes, torch
sys.path.insert(0, '.')
import numpy as np
from dualpixelface_amd import ops, _lib
SH = {'hg32': (4, 32, 8, 256, 384, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)), 'fe32': (4, 32, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
      'hg64': (4, 64, 4, 128, 192, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)), 'fe32q': (4, 32, 1, 256, 384, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
      'fe96_32': (4, 96, 1, 512, 768, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1))}
for nm in sys.argv[1:] or ['fe32', 'hg32']:
    N, C, D, H, W, K, ks, st, pd = SH[nm]
    x = torch.randn(N, C, D, H, W, device='cuda'); w = torch.randn(K, C, *ks, device='cuda') * 0.1
    for _ in range(3):
        y = ops.ConvFn.apply(x, w, None, st, pd, (1, 1, 1))
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); y = ops.ConvFn.apply(x, w, None, st, pd, (1, 1, 1)); t1.record(); torch.cuda.synchronize()
    nb = 16384
    buf = (ctypes.c_ulonglong * (8 * nb))()
    assert _lib.lib().cdll.dpf_debug_x9_stamps(buf, nb) == 0
    a = np.array(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    a = a[a[:, 3] > a[:, 0]]
    a = a[a[:, 7] == a[0, 7]]
    pro, loop, epi, tot = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 3] - a[:, 0]
    real = (a[:, 5] - a[:, 4]) * 10.0   # ns (100 MHz)
    q = lambda v: '%7.0f %7.0f %7.0f' % tuple(np.percentile(v, [10, 50, 90]))
    print('%s: %d workgroups, kernel %.3f ms; clocks p10/p50/p90  prologue %s | chunk loop %s | epilogue %s | total %s | ns %s | clock %.2f GHz' % (
        nm, len(a), t0.elapsed_time(t1), q(pro), q(loop), q(epi), q(tot), q(real), np.median(tot / np.maximum(real, 1))))
    span = (a[:, 5].max() - a[:, 4].min()) * 10e-6
    print('   first start -> last end %.3f ms; sum of workgroup times / (512 slots) = %.3f ms' % (span, real.sum() * 1e-6 / 512))
