cd /root/repo
./tools/lean_probe
python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from dualpixelface_amd import ops
dev='cuda'
def bench(C, off, tag):
    x = torch.randn(4, C, 4, 256, 384, device=dev); w = torch.randn(64, C, 3, 3, 3, device=dev) * 0.05; b = torch.zeros(64, device=dev)
    tf = []
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.deform_conv_forward_raw(x, w, b, off, (1, 1, 1), (1, 1, 1), (1, 1, 1))
        torch.cuda.synchronize(); tf.append((time.perf_counter() - t0) * 1e3)
    print('C=%d %-28s fwd min %.2f ms' % (C, tag, min(tf[1:])))
torch.manual_seed(0)
for C in (64, 35):
    for sig in (0.0, 0.3, 0.75, 1.3):
        bench(C, torch.randn(4, 81, 4, 256, 384, device=dev) * sig, 'iid sigma %.2f' % sig)
    sm = torch.randn(4, 81, 4, 256, 1, device=dev).expand(4, 81, 4, 256, 384).contiguous() * 1.3
    bench(C, sm, 'row-constant sigma 1.3')
    sm = torch.nn.functional.interpolate(torch.randn(4, 81, 4, 32, 48, device=dev).view(4 * 81, 1, 4, 32, 48), size=(4, 256, 384), mode='trilinear').view(4, 81, 4, 256, 384).contiguous() * 2.0
    bench(C, sm, 'smooth (8x upsampled) ~1.3')
PY
