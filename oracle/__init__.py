"""ORACLE -- CPU restatements of the reference's StereoDPNet path (test infrastructure only).

Nothing in ``dualpixelface_amd`` may import this package; see oracle/stereodpnet.py and
oracle/dcn3d.py for the per-function reference citations and the parity-pinning status.
"""
import json
import os

import torch

_KEYS = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'state_dict_keys.json')


def recipe_state(requires_grad=True, dtype=torch.float32, keys_file=None):
    """Oracle ``state`` dict filled by the shared recipe (dualpixelface_amd/recipe.py); ``keys_file``: another model's
    {state_dict key: shape} fixture (default: StereoDPNet's)."""
    from dualpixelface_amd.recipe import recipe_tensor, SKIP_SUFFIXES
    shapes = json.load(open(keys_file or _KEYS))
    st = {}
    for k, shp in shapes.items():
        if k.endswith('num_batches_tracked'):
            st[k] = torch.zeros((), dtype=torch.long)
            continue
        if k.endswith(SKIP_SUFFIXES):
            continue
        t = recipe_tensor(k, torch.empty(shp, dtype=torch.float32)).to(dtype)
        if requires_grad and not k.endswith(('running_mean', 'running_var')):
            t.requires_grad_()
        st[k] = t
    return st
