#!/usr/bin/env python3
"""Golden vectors for the PSMNet plugin (BASELINE configs[3]) by IMPORTING THE REFERENCE's src/model/psmnet (build container
only; inputs are recipe.synthetic_batch(2, 256, 256, seed=7) and are not stored; shims of make_golden.py: pytorch_lightning stub, Tensor.cuda -> identity, metric_type = []).  Run from the repo root:
    python tests/golden/make_golden_psmnet.py
256x256 is the smallest input the reference accepts (its 64x64 average pool runs on the quarter-resolution map)."""
import importlib.util
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
spec = importlib.util.spec_from_file_location('make_golden', str(HERE / 'make_golden.py'))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch  # noqa: E402

GRAD_KEYS = ['feature_extraction.firstconv.0.0.weight', 'feature_extraction.layer2.0.downsample.0.weight',
             'feature_extraction.layer4.2.conv2.0.weight', 'feature_extraction.branch1.1.0.weight',
             'feature_extraction.lastconv.2.weight', 'aggregation.dres0.0.0.weight', 'aggregation.classif3.2.weight']


def main():
    mg.install_shims()
    torch.manual_seed(1)
    model, opt = mg.build_reference('psmnet')
    fill_by_recipe(model)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    json.dump(keys, open(HERE / 'psmnet_state_dict_keys.json', 'w'), indent=0)
    out = {}
    for tag, train in (('train', True), ('eval', False)):
        fill_by_recipe(model)
        model.train(train)
        batch = synthetic_batch(2, 256, 256, seed=7)      # 2 samples: BatchNorm of the 1x1 SPP branch needs > 1 value per channel
        for p in model.parameters():
            p.grad = None
        res = model(batch)
        if train:
            res['final_loss'].backward()
            pd = dict(model.named_parameters())
            for k in GRAD_KEYS:
                if pd[k].numel() <= 8192:
                    out['grad::' + k] = mg.f32(pd[k].grad)
                out['gradcs::' + k] = mg.checksum(pd[k].grad)
            out['smoothL1_loss'] = mg.f32(res['smoothL1_loss'])
            out['final_loss'] = mg.f32(res['final_loss'])
            sd = model.state_dict()
            out['post::feature_extraction.branch1.1.1.running_mean'] = mg.f32(sd['feature_extraction.branch1.1.1.running_mean']).copy()   # not a view of the live buffer
        out[tag + '_pred_depth_s2'] = mg.f32(res['pred_depth'][:, :, ::2, ::2])
        out[tag + '_pred_depth_cs'] = mg.checksum(res['pred_depth'])
        out[tag + '_ref_feature'] = mg.f32(res['ref_feature'])
        out[tag + '_prob_cs'] = mg.checksum(res['prob_depth'])
    np.savez_compressed(HERE / 'psmnet_256x256_b2.npz', **out)
    print('keys', len(keys), 'loss', float(out['final_loss']), {k: v.shape for k, v in out.items() if 'pred_depth_s2' in k})


if __name__ == '__main__':
    main()
