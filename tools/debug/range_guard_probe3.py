import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from dualpixelface_amd import ops
from dualpixelface_amd._lib import lib
from test_gpu_ops import rnd
N, C, K, D, H, W = 1, 32, 32, 4, 24, 64
pad = (1, 1, 1); one = (1, 1, 1)
g = torch.Generator().manual_seed(5)
def run(a, w, tag, guards=(1,)):
    ref = F.conv3d(a.double(), w.double(), None, 1, pad); den = F.conv3d(a.abs().double(), w.abs().double(), None, 1, pad)
    out = []
    for path, guard in [(0, 1)] + [(2, gd) for gd in guards]:
        lib().call('dpf_set_f32_matrix_path', path); lib().call('dpf_debug_set_range_guard', guard)
        got = ops.ConvFn.apply(a.cuda(), w.cuda(), None, one, pad, one).double().cpu()
        e = (got - ref).abs() / den
        err = e.amax(dim=(0, 1, 2, 3))
        out.append('p%d g%d: big %.2e tiny max %.2e median %.2e' % (path, guard, err[2:5].max().item(), err[10:60].max().item(), e[..., 10:60].median().item()))
    lib().call('dpf_debug_set_range_guard', 1)
    print(tag, '|', ' | '.join(out))
sign = lambda *s: (torch.randint(0, 2, s, generator=g) * 2 - 1).float()
big = (2 + 1.9 * torch.rand(N, C, D, H, 6, generator=g)) * sign(N, C, D, H, 6)
bigx = (torch.randint(1024, 2000, (N, C, D, H, 6), generator=g).float() / 512) * sign(N, C, D, H, 6)     # f16-exact after scaling
w = rnd(K, C, 3, 3, 3, seed=311, scale=0.1)
tiny = torch.randn(N, C, D, H, W - 6, generator=g) * 2.0 ** -34
a = torch.cat([big, tiny], dim=4)
run(a, w, 'E0 baseline          ')
a1 = a.clone(); a1[:, 4:] = 0
run(a1, w, 'E1 one active chunk  ')
run(torch.cat([bigx, tiny], dim=4), w, 'E3 big f16-exact     ')
plain = torch.randn(N, C, D, H, W, generator=g)
run(plain, w, 'E4 plain gauss, forced passes', guards=(1, 0))
run(plain.abs(), w.abs(), 'E4 plain positive, forced passes', guards=(1, 0))
run(torch.cat([big, tiny.abs()], dim=4), w.abs(), 'E5 positive tiny, w>0')
print('---- where are the worst elements (E0, path 2)?')
ref = F.conv3d(a.double(), w.double(), None, 1, pad); den = F.conv3d(a.abs().double(), w.abs().double(), None, 1, pad)
lib().call('dpf_set_f32_matrix_path', 2)
got = ops.ConvFn.apply(a.cuda(), w.cuda(), None, one, pad, one).double().cpu()
e = ((got - ref).abs() / den)[0]
e[..., :10] = 0
top = torch.topk(e.flatten(), 25)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    k, r = divmod(i, D * H * W); d, r = divmod(r, H * W); h, x = divmod(r, W)
    print('err %.2e k %d d %d h %d w %d   |ref|/den %.3f' % (v, k, d, h, x, (ref[0, k, d, h, x].abs() / den[0, k, d, h, x]).item()))
print('per-h max:', ' '.join('%.1e' % v for v in e.amax(dim=(0, 1, 3)).tolist()))
print('per-d max:', ' '.join('%.1e' % v for v in e.amax(dim=(0, 2, 3)).tolist()))
print('per-k max:', ' '.join('%.1e' % v for v in e.amax(dim=(1, 2, 3)).tolist()))
