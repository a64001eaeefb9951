"""Stand-alone timing of the deformable-conv kernels at the StereoDPNet shapes (B=4, 4x256x384 voxels).  argv: [fwd|all] [C ...]"""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
dev = 'cuda'
mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
sel = [int(a) for a in sys.argv[2:]] or [35, 64]
for C, sig in ((35, 1.3), (64, 0.75)):
    if C not in sel:
        continue
    torch.manual_seed(0)
    x = torch.randn(4, C, 4, 256, 384, device=dev)
    off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig
    w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(64, device=dev)
    go = torch.randn(4, 64, 4, 256, 384, device=dev)
    tf, tb = [], []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if mode == 'all':
            g = ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tf.append((t1 - t0) * 1e3); tb.append((t2 - t1) * 1e3)
    print('C=%d sigma=%.2f  fwd min %.2f med %.2f ms   bwd(all) min %.2f med %.2f ms' % (C, sig, min(tf[1:]), sorted(tf[1:])[2], min(tb[1:]), sorted(tb[1:])[2]))
