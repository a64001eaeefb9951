#!/usr/bin/env python3
"""Golden vectors for the NNet plugin by IMPORTING THE REFERENCE's src/model/nnet (build container
only; inputs are recipe.synthetic_batch(2, 256, 256, seed=11) and are not stored; shims of make_golden.py: pytorch_lightning stub,
Tensor.cuda -> identity, metric_type = []).  Run from the repo root:
    python tests/golden/make_golden_nnet.py
256x256 is the smallest input the reference accepts (its 64x64 average pool runs on the quarter-resolution map)."""
import importlib.util
import json
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
spec = importlib.util.spec_from_file_location('make_golden', str(HERE / 'make_golden.py'))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch  # noqa: E402

GRAD_KEYS = ['feature_extraction.firstconv.0.0.weight', 'feature_extraction.branch1.1.0.weight', 'feature_extraction.lastconv.2.weight',
             'convs.0.0.weight', 'convs.4.0.weight', 'convs.6.0.weight', 'dres0.0.0.weight', 'dres3.2.0.weight', 'classify.2.weight',
             'normal_module.wc0.0.0.weight', 'normal_module.pool2.0.0.weight', 'normal_module.pool3.0.1.weight',
             'normal_module.n_convs.4.0.weight', 'normal_module.n_convs.6.0.weight']


def main():
    mg.install_shims()
    torch.manual_seed(1)
    model, opt = mg.build_reference('nnet')
    fill_by_recipe(model)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    out = {}
    cap = {}
    model.normal_module.wc0.register_forward_pre_hook(lambda m, inp: cap.__setitem__('wc', inp[0]))
    model.normal_module.pool3.register_forward_hook(lambda m, inp, o: cap.__setitem__('pool3', o))
    model.classify.register_forward_hook(lambda m, inp, o: cap.__setitem__('costs', o))
    for tag, train in (('train', True), ('eval', False)):
        fill_by_recipe(model)
        model.train(train)
        batch = synthetic_batch(2, 256, 256, seed=11)
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
            for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
                out[k] = mg.f32(res[k])
            sd = model.state_dict()
            out['post::normal_module.pool1.0.1.running_mean'] = mg.f32(sd['normal_module.pool1.0.1.running_mean']).copy()
            out['post::dres2.0.1.running_var'] = mg.f32(sd['dres2.0.1.running_var']).copy()
            out['train_xyz_s'] = mg.f32(cap['wc'][:, :3, :, ::4, ::4])
            out['train_wc_cs'] = mg.checksum(cap['wc'])
            out['train_pool3_s'] = mg.f32(cap['pool3'][:, :, :, ::4, ::4])
            out['train_costs_s'] = mg.f32(cap['costs'][:, :, :, ::2, ::2])
        out[tag + '_pred_depth_s2'] = mg.f32(res['pred_depth'][:, :, ::2, ::2])
        out[tag + '_pred_depth_cs'] = mg.checksum(res['pred_depth'])
        out[tag + '_pred_normal_s2'] = mg.f32(res['pred_normal'][:, :, :, ::2, ::2])
        out[tag + '_pred_normal_cs'] = mg.checksum(res['pred_normal'])
        out[tag + '_ref_feature'] = mg.f32(res['ref_feature'])
        out[tag + '_prob_cs'] = mg.checksum(res['prob_depth'])
    keys_after = {k: list(v.shape) for k, v in model.state_dict().items()}      # includes the lazily registered normal_module.grid
    json.dump(keys_after, open(HERE / 'nnet_state_dict_keys.json', 'w'), indent=0)
    np.savez_compressed(HERE / 'nnet_256x256_b2.npz', **out)
    print('keys', len(keys), len(keys_after), 'loss', float(out['final_loss']), float(out['smoothL1_loss']), float(out['cosine_loss']))


if __name__ == '__main__':
    main()
