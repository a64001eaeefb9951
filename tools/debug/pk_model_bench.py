"""grad_input kernel at the bench model's offset statistics (layer 1: sigma 1.3 with a heavy tail, layer 2: 0.75)."""
import sys, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
torch.manual_seed(0)
dev = 'cuda'
for C, sig in ((35, 1.05), (64, 0.63)):
    x = torch.randn(4, C, 4, 256, 384, device=dev)
    off = torch.randn(4, 81, 4, 256, 384, device=dev) * sig * 1.25
    off = off * (1 + 2.0 * (torch.rand_like(off) < 0.02))        # 2 % of the offsets three times as large (the model's tail)
    w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05
    b = torch.zeros(64, device=dev)
    go = torch.randn(4, 64, 4, 256, 384, device=dev)
    for _ in range(3):
        ops.deform_conv_backward_raw(x, w, b, off, go, (1, 1, 1), (1, 1, 1), (1, 1, 1))
torch.cuda.synchronize()
