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
def run(a, w, tag):
    ref = F.conv3d(a.double(), w.double(), None, 1, pad); den = F.conv3d(a.abs().double(), w.abs().double(), None, 1, pad)
    out = []
    for path in (0, 2):
        lib().call('dpf_set_f32_matrix_path', path)
        got = ops.ConvFn.apply(a.cuda(), w.cuda(), None, one, pad, one).double().cpu()
        err = ((got - ref).abs() / den).amax(dim=(0, 1, 2, 3))
        out.append('path %d: big cols %.2e  tiny cols %.2e' % (path, err[2:5].max().item(), err[10:60].max().item()))
    print(tag, '|', ' | '.join(out))
# columns 0..5: big band, values in [2, 3.9] (scale 2^13); the rest: tiny band
big = (2 + 1.9 * torch.rand(N, C, D, H, 6, generator=g)) * (torch.randint(0, 2, (N, C, D, H, 6), generator=g) * 2 - 1)
w_gauss = rnd(K, C, 3, 3, 3, seed=311, scale=0.1)
w_exact = torch.randint(-32, 33, (K, C, 3, 3, 3), generator=g).float() / 64
for d in (34, 30, 38):
    tiny_gauss = torch.randn(N, C, D, H, W - 6, generator=g) * 2.0 ** -d
    m = torch.randint(-8, 9, (N, C, D, H, W - 6), generator=g).double()
    t = torch.randint(-1023, 1024, (N, C, D, H, W - 6), generator=g).double()
    tiny_exact = ((m * 2.0 ** -24 + t * 2.0 ** -35) * 2.0 ** -13 * 2.0 ** (34 - d)).float()      # hi0 = m 2^-24, remainder = t 2^-35: 11 bits
    for wn, w in (('w gauss', w_gauss), ('w f16-exact', w_exact)):
        for xn, tiny in (('x gauss', tiny_gauss), ('x two-piece', tiny_exact)):
            run(torch.cat([big, tiny], dim=4), w, 'd=%d %s %s' % (d, wn, xn))
