import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
for (N, C, K, D, H, W) in ((4, 32, 32, 1, 256, 384), (4, 32, 32, 3, 256, 384), (4, 64, 64, 1, 128, 192), (4, 128, 128, 1, 64, 96), (4, 64, 32, 1, 128, 192)):
    x = torch.randn(N, C, D, H, W, device='cuda'); g = torch.randn(N, K, D, H, W, device='cuda')
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ops._conv_wgrad_raw(g, x, (K, C, 1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10)
    t = min(ts[1:]); nb = 4.0 * (x.numel() + g.numel())
    print('pw wgrad N%d C%d K%d %dx%dx%d: %.1f us  %.2f TB/s' % (N, C, K, D, H, W, t * 1e6, nb / t / 1e12), flush=True)
