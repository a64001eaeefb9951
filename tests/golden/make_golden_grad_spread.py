#!/usr/bin/env python3
"""The imported reference's OWN fp32 gradient scatter, per parameter tensor (build container only; needs /root/reference).

    python tests/golden/make_golden_grad_spread.py      ->  tests/golden/grad_spread.npz

For each gradient fixture of make_golden.py (train_32x48_b2, train_64x96_b1, train_128x128_b2: same recipe weights, same synthetic batch)
the reference model (same shims as make_golden.py) runs forward + loss + backward at 1, 2, 4 and 8 intra-op threads -- four equally
valid fp32 summation orders of the SAME program -- and once in fp64 (model.double(), double inputs).  Stored per fixture:

  names                         every parameter that receives a gradient
  spread[n]                     max over the 6 thread pairs of |g_a - g_b|_2 / |g_8|_2       (the reference's self-spread, fp32 vs fp32)
  d64[n, 4]                     |g_t - g_fp64|_2 / |g_fp64|_2 for t = 1, 2, 4, 8 threads     (the reference's fp32 distance to fp64)
  norm64[n]                     |g_fp64|_2
  check8[n]                     |g_8 - fixture checksum| consistency: sum g^2 of the 8-thread run (must equal e2e_<tag>.npz's grad_cs[:, 2])
  grad64::<name>                the fp64 gradient of the 10 tensors whose fp32 gradient e2e_<tag>.npz stores in full

tests/test_gpu_e2e.py derives its gradient budgets from these numbers (budget = K_SPREAD x spread, fp64 bound = K_FP64 x max_t d64): the
tolerance of the HIP path is a multiple of what the reference itself does when only its thread count changes, not an envelope of the HIP
kernels' own variants.  The 8-thread run is bit-identical to the committed e2e fixtures (asserted below), so the spreads are spreads AROUND
the fixture the tests compare against.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg    # noqa: E402

from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch  # noqa: E402

FIXTURES = [('train_32x48_b2', 2, 32, 48, 'bern'), ('train_64x96_b1', 1, 64, 96, 'ones'), ('train_128x128_b2', 2, 128, 128, 'bern')]
THREADS = (1, 2, 4, 8)


def run(B, H, W, mask_mode, threads, double):
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    model, opt = mg.build_reference()
    fill_by_recipe(model)
    model.train(True)
    batch = synthetic_batch(B, H, W, seed=0, mask_mode=mask_mode)
    real_float = torch.Tensor.float
    if double:
        # the reference hard-casts to fp32 inside its deformable-conv wrapper and its phase shift (deform_conv.py:55-57,91-94,
        # deform_conv_func.py:27-28, asm.py:112: `.float()`); for the fp64 leg ONLY that cast is made a cast to double (a shim on torch, the
        # reference files are untouched), so that the whole program runs in fp64
        model.double()
        batch = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
        torch.Tensor.float = lambda self, *a, **k: self.double()
    try:
        res = model(batch)
        res['final_loss'].backward()
    finally:
        torch.Tensor.float = real_float
    return {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None}, float(res['final_loss'])


def main():
    mg.install_shims()
    out = {}
    for tag, B, H, W, mode in FIXTURES:
        fix = np.load(mg.OUT / ('e2e_%s.npz' % tag))
        runs = {}
        for t in THREADS:
            runs[t], loss = run(B, H, W, mode, t, False)
            print(tag, 'threads', t, 'loss %.7f' % loss, flush=True)
        g64, loss64 = run(B, H, W, mode, 8, True)
        print(tag, 'fp64 loss %.10f' % loss64, flush=True)
        names = [str(n) for n in fix['grad_names']]
        assert set(names) == set(runs[8].keys()), 'gradient set differs from the fixture'
        # the 8-thread run IS the committed fixture (make_golden.py runs at 8 threads)
        for n, c in zip(names, fix['grad_cs']):
            assert abs((runs[8][n] ** 2).sum().item() - c[2]) <= 1e-12 * max(abs(c[2]), 1e-30), (tag, n)
        spread, d64, norm64, check8 = [], [], [], []
        for n in names:
            ref = runs[8][n]
            nr = max(ref.norm().item(), 1e-30)
            s = 0.0
            for i, a in enumerate(THREADS):
                for b in THREADS[i + 1:]:
                    s = max(s, (runs[a][n] - runs[b][n]).norm().item() / nr)
            spread.append(s)
            n64 = max(g64[n].norm().item(), 1e-30)
            d64.append([(runs[t][n] - g64[n]).norm().item() / n64 for t in THREADS])
            norm64.append(g64[n].norm().item())
            check8.append((ref ** 2).sum().item())
        out[tag + '/names'] = np.array(names)
        out[tag + '/spread'] = np.array(spread, dtype=np.float64)
        out[tag + '/d64'] = np.array(d64, dtype=np.float64)
        out[tag + '/norm64'] = np.array(norm64, dtype=np.float64)
        out[tag + '/check8'] = np.array(check8, dtype=np.float64)
        out[tag + '/loss64'] = np.float64(loss64)
        for k in fix.files:
            if k.startswith('grad::'):
                out[tag + '/grad64::' + k[6:]] = g64[k[6:]].numpy()
    out['threads'] = np.array(THREADS)
    np.savez_compressed(mg.OUT / 'grad_spread.npz', **out)
    print('wrote grad_spread.npz')


if __name__ == '__main__':
    main()
