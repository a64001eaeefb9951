"""Per-band error of the f16-component conv path against fp64 (diagnostic for the range guard)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch, torch.nn.functional as F
from dualpixelface_amd import ops
from dualpixelface_amd._lib import lib
from test_gpu_ops import _spread, rnd, RANGE_EXPS
N, C, K, D, H, W = 1, 32, 32, 4, 24, 64
ks = (3, 3, 3); pad = (1, 1, 1); one = (1, 1, 1)
for positive in (False, True):
    a = rnd(N, C, D, H, W, seed=310); w = rnd(K, C, *ks, seed=311, scale=0.1)
    if positive: a, w = a.abs(), w.abs()
    a = _spread(a, 'wbands', 312)
    ref = F.conv3d(a.double(), w.double(), None, 1, pad); den = F.conv3d(a.abs().double(), w.abs().double(), None, 1, pad)
    for path, guard in ((0, 1), (2, 1), (2, 2), (2, 3), (2, 4)):
        lib().call('dpf_set_f32_matrix_path', path); lib().call('dpf_debug_set_range_guard', guard)
        got = ops.ConvFn.apply(a.cuda(), w.cuda(), None, one, pad, one).double().cpu()
        err = ((got - ref).abs() / den)
        percol = err.amax(dim=(0, 1, 2, 3))
        print('positive', positive, 'path', path, 'guard', guard, 'max %.3e' % err.max().item())
        print('   per column:', ' '.join('%.1e' % v for v in percol.tolist()[12:32]))
lib().call('dpf_debug_set_range_guard', 1)
