"""Caches the fp64 CPU oracle's gradients at BASELINE configs[1]'s shape (1 x 512 x 768, recipe weights, synthetic_batch seed 21):
tests/golden/c2_fp64_grads.npz = full gradients of 12 parameters spread over the network, {sum g^2} of every parameter gradient, the
losses and the ANM level selection.  The oracle (oracle/stereodpnet.py) is pinned to the imported reference at 32x48 ... 128x128 by
make_golden.py / tests/test_oracle_golden.py; this file only saves the ~1-3 minutes its fp64 run takes at the production tile sizes.
    python tests/golden/make_golden_c2_fp64.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dualpixelface_amd.recipe import synthetic_batch     # noqa: E402
from oracle import recipe_state                          # noqa: E402
from oracle.stereodpnet import StereoDPNetOracle         # noqa: E402

FULL = ['aggregation.classif3.2.weight', 'cost_volume.attention_layer.mask_convs.0.weight', 'cost_volume.attention_layer.normalize.weight',
        'normal_estimator.deform_conv1.bias', 'normal_estimator.deform_conv1.conv_offset.bias', 'normal_estimator.deform_conv2.weight',
        'normal_estimator.n_convs.5.0.weight', 'feature_extraction.firstconv.0.0.weight', 'feature_extraction.block1.prelu.weight',
        'feature_extraction.fpn.inner_blocks.0.bias', 'aggregation.dres2.conv6.0.weight', 'aggregation.dres0.0.0.weight']


def main():
    batch = synthetic_batch(1, 512, 768, seed=21, mask_mode='bern')
    st = recipe_state(dtype=torch.float64)
    orc = StereoDPNetOracle(st, training=True)
    res = orc.forward({k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()})
    res['final_loss'].backward()
    out = {'final_loss': np.float64(res['final_loss'].item()), 'smoothL1_loss': np.float64(res['smoothL1_loss'].item()),
           'cosine_loss': np.float64(res['cosine_loss'].item()), 'anm_idx': orc.taps['anm_idx'].to(torch.uint8).numpy()}
    names, cs = [], []
    for k, t in st.items():
        if getattr(t, 'grad', None) is not None:
            names.append(k)
            cs.append(float((t.grad ** 2).sum()))
    out['grad_names'] = np.array(names)
    out['grad_sumsq'] = np.array(cs, dtype=np.float64)
    for k in FULL:
        out['grad::' + k] = st[k].grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c2_fp64_grads.npz'), **out)
    print('saved', len(names), 'checksums,', len(FULL), 'full gradients')


if __name__ == '__main__':
    main()
