"""Stand-alone timing of the deformable-conv kernels at the StereoDPNet shapes (B=4, 4x256x384 voxels)."""
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
dev = 'cuda'
for C, sig in ((35, 1.3), (64, 0.75)):
    torch.manual_seed(0)
    x = torch.randn(4, C, 4, 256, 384, device=dev)
    off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig
    w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(64, device=dev)
    go = torch.randn(4, 64, 4, 256, 384, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        g = ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print('C=%d sigma=%.2f  fwd %.2f ms   bwd(all) %.2f ms' % (C, sig, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
