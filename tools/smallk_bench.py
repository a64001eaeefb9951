import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
import torch.nn.functional as F
x = torch.randn(4, 32, 8, 256, 384, device='cuda'); w = torch.randn(1, 32, 3, 3, 3, device='cuda') * 0.1
y = ops.conv3d(x, w, None, 1, 1, 1)
print('err', float((y - F.conv3d(x, w, None, 1, 1)).abs().max()))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.conv3d(x, w, None, 1, 1, 1)
    torch.cuda.synchronize(); print('fwd ms', (time.perf_counter() - t0) / 10 * 1e3)
x2 = torch.randn(16, 32, 1, 256, 384, device='cuda'); w2 = torch.randn(3, 32, 1, 3, 3, device='cuda') * 0.1
y2 = ops.conv3d(x2, w2, None, 1, (0, 1, 1), 1)
print('err2', float((y2 - F.conv3d(x2, w2, None, 1, (0, 1, 1))).abs().max()))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ops.conv3d(x2, w2, None, 1, (0, 1, 1), 1)
torch.cuda.synchronize(); print('fwd K3 ms', (time.perf_counter() - t0) / 10 * 1e3)
