#!/usr/bin/env python3
"""Golden vectors for the *fix mode* of the cost volume (`asm_grid_cache_compat = false`): per-level shifts, i.e. fractional
Fourier-phase shifts (src/module/asm/asm.py:59-75,112-125), which the reference as written never reaches because its shift-grid
cache is keyed on nothing.  Produced by IMPORTING THE REFERENCE (build container only):

    python tests/golden/make_golden_fixmode.py   ->  tests/golden/shift_fractional.npz, tests/golden/e2e_fixmode_train_32x48_b2.npz

  * shift_fractional.npz: `subpixel_shift.forward(fea, delta, dir)` of a FRESH module instance per call (so its cache holds the
    requested delta) for fractional and integer deltas, both directions -- the reference's own arithmetic, through the
    rfft / irfft shims of make_golden.py;
  * e2e_fixmode_*.npz: the whole reference model with the four cache attributes of its `shifting_layer` cleared before every call
    (a forward pre-hook set from here; no reference source is modified) -- "the reference with Q1 fixed".
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
spec = importlib.util.spec_from_file_location('make_golden', str(HERE / 'make_golden.py'))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def gen_shift():
    model, opt = mg.build_reference()
    cls = type(model.cost_volume.shifting_layer)
    out = {}
    g = torch.Generator().manual_seed(7)
    cases = [(2, 3, 16, 24), (1, 2, 8, 12), (1, 1, 32, 20)]
    deltas = [0.5, -0.25, 1.75, -1.5, 3.0, 0.125]
    out['deltas'] = np.array(deltas)
    for ci, shape in enumerate(cases):
        fea = torch.randn(*shape, generator=g)
        out['fea%d' % ci] = fea.numpy()
        for di, delta in enumerate(deltas):
            for direction in ('forward', 'backward'):
                layer = cls(opt)                                      # fresh instance: empty caches
                near, bil, ph = layer(fea, delta, direction)
                key = 'c%d_d%d_%s_' % (ci, di, direction)
                out[key + 'nearest'], out[key + 'bilinear'], out[key + 'phase'] = (mg.f32(t[..., 0]) for t in (near, bil, ph))
    np.savez_compressed(mg.OUT / 'shift_fractional.npz', **out)
    print('shift_fractional.npz', len(out))


def clear_shift_cache(model):
    layer = model.cost_volume.shifting_layer

    def pre(mod, inp):
        mod.basic_grid_forward = mod.basic_grid_backward = mod.phase_grid_forward = mod.phase_grid_backward = None
    layer.register_forward_pre_hook(pre)


if __name__ == '__main__':
    mg.install_shims()
    torch.set_num_threads(8)
    gen_shift()
    mg.gen_e2e('fixmode_train_32x48_b2', 2, 32, 48, True, 'bern', stages=True, prepare=clear_shift_cache)
    mg.gen_e2e('fixmode_eval_32x48_b2', 2, 32, 48, False, 'ones', stages=False, prepare=clear_shift_cache)
