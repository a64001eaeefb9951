"""Deterministic parameter recipe shared by the golden-vector generator and the tests.

Every ``state_dict`` entry is filled from ``torch.Generator().manual_seed(crc32(key))`` so that the
reference model (imported only in the build container, tests/golden/make_golden.py) and this
repo's model hold identical 3.67 M parameters without shipping a 14.7 MB checkpoint
(SURVEY.md section 8c).  Frozen geometry parameters (``costrange``, lazily registered ``grid``,
normal_module.py:76-78,91-99) are left untouched.
"""
import math
import zlib

import torch

SKIP_SUFFIXES = ('costrange', '.grid', 'num_batches_tracked')


def _gen(key):
    return torch.Generator().manual_seed(zlib.crc32(key.encode()))


def recipe_tensor(key, ref):
    """Value for ``key`` with the shape/dtype of ``ref`` (a tensor)."""
    g = _gen(key)
    shape = tuple(ref.shape)
    if key.endswith('running_var'):
        v = torch.rand(shape, generator=g) + 0.5
    elif key.endswith('running_mean'):
        v = torch.randn(shape, generator=g) * 0.1
    elif ref.dim() <= 1 and key.endswith('weight'):
        if ref.numel() == 1:                      # PReLU slope
            v = torch.rand(shape, generator=g) * 0.28 + 0.02
        else:                                     # BatchNorm / InstanceNorm scale
            v = torch.rand(shape, generator=g) + 0.5
    elif ref.dim() <= 1:                          # biases
        v = torch.randn(shape, generator=g) * 0.1
    else:                                         # conv / deconv / DCN kernels
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        v = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
    return v.to(ref.dtype)


@torch.no_grad()
def fill_by_recipe(module):
    """In-place fill of every parameter and buffer of ``module`` (sorted key order)."""
    sd = module.state_dict()
    for key in sorted(sd.keys()):
        if key.endswith(SKIP_SUFFIXES):
            continue
        sd[key].copy_(recipe_tensor(key, sd[key]))
    return module


def synthetic_batch(B, H, W, seed=0, mask_mode='ones', device='cpu'):
    """Synthetic dual-pixel batch with the FaceDP loader's keys/shapes/dtypes
    (dataloader/FaceDP/loader.py:149-155; abvalue = [b, a], path_reader.py:26,203)."""
    g = torch.Generator().manual_seed(1000 + seed)
    ab = torch.tensor([32.98, -26996.49])
    left = torch.randn(B, 3, H, W, generator=g)
    right = torch.randn(B, 3, H, W, generator=g)
    disp = torch.rand(B, H, W, generator=g) * 8.0 - 2.0
    depth = ab[1] / (disp - ab[0])
    idepth = depth.amax(dim=(1, 2), keepdim=True) / depth
    normal = torch.randn(B, 3, H, W, generator=g)
    normal = normal / normal.norm(dim=1, keepdim=True).clamp_min(1e-6)
    if mask_mode == 'ones':
        mask = torch.ones(B, H, W)
    else:
        mask = (torch.rand(B, H, W, generator=g) < 0.8).float()
    K = torch.tensor([[5000.0, 0.0, W / 2.0], [0.0, 5000.0, H / 2.0], [0.0, 0.0, 1.0]]).repeat(B, 1, 1)
    batch = {'left': left, 'right': right, 'disp': disp, 'depth': depth, 'idepth': idepth,
             'mask': mask, 'normal': normal, 'K': K, 'abvalue': ab.repeat(B, 1)}
    return {k: v.to(device) for k, v in batch.items()}
