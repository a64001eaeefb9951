#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (build container only).

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (read-only); nothing from it is copied -- only inputs/outputs (data) are
saved.  The GPU box never runs this script.

Shims (all out of the reference tree):
  1. pytorch_lightning.LightningModule  -> nn.Module + no-op save_hyperparameters/log
  2. torchvision.ops.FeaturePyramidNetwork -> restated from torchvision 0.6/0.7's published
     semantics (1x1 lateral + 3x3 output convs with bias, nearest top-down) -- third-party code
     absent from /root/reference => parity unpinned at that boundary (6 plain convs)
  3. torch.rfft / torch.irfft (removed APIs) -> torch.fft forms
  4. Tensor.cuda -> identity; grad-requiring leaves are cloned; same for type_as
  5. module ``DCN`` -> oracle/dcn3d.py (the reference CUDA op has no CPU build; outputs that pass
     through it are flagged *derived*)
  6. metric_type = [] (TensorFlow/texttable absent; metrics are off-path)
"""
import json
import os
import sys
import types
import zlib
from collections import OrderedDict
from pathlib import Path
from runpy import run_path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parents[2]
REF = Path('/root/reference')
OUT = REPO / 'tests' / 'golden'
sys.path.insert(0, str(REPO))

from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch  # noqa: E402
from oracle import dcn3d as oracle_dcn  # noqa: E402


# ----------------------------------------------------------------------------- shims
def install_shims():
    pl = types.ModuleType('pytorch_lightning')

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    sys.modules['pytorch_lightning'] = pl

    tv = types.ModuleType('torchvision')
    tvo = types.ModuleType('torchvision.ops')

    class FeaturePyramidNetwork(nn.Module):
        def __init__(self, in_channels_list, out_channels, extra_blocks=None):
            super().__init__()
            self.inner_blocks = nn.ModuleList()
            self.layer_blocks = nn.ModuleList()
            for c in in_channels_list:
                self.inner_blocks.append(nn.Conv2d(c, out_channels, 1))
                self.layer_blocks.append(nn.Conv2d(out_channels, out_channels, 3, padding=1))
            for m in self.children():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_uniform_(m.weight, a=1)
                    nn.init.constant_(m.bias, 0)

        def forward(self, x):
            names = list(x.keys())
            xs = list(x.values())
            last = self.inner_blocks[-1](xs[-1])
            res = [self.layer_blocks[-1](last)]
            for i in range(len(xs) - 2, -1, -1):
                lat = self.inner_blocks[i](xs[i])
                last = lat + F.interpolate(last, size=lat.shape[-2:], mode='nearest')
                res.insert(0, self.layer_blocks[i](last))
            return OrderedDict(zip(names, res))

    tvo.FeaturePyramidNetwork = FeaturePyramidNetwork
    tv.ops = tvo
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.ops'] = tvo

    def rfft(x, ndim, onesided=False):
        assert ndim == 2 and not onesided
        return torch.view_as_real(torch.fft.fft2(x))

    def irfft(c, ndim, onesided=False):
        assert ndim == 2 and not onesided
        Hh, Ww = c.shape[-3], c.shape[-2]
        cc = torch.view_as_complex(c.contiguous())
        return torch.fft.irfft2(cc[..., :Ww // 2 + 1], s=(Hh, Ww))

    torch.rfft = rfft
    torch.irfft = irfft

    def cuda(self, *a, **k):
        return self.clone() if (self.is_leaf and self.requires_grad) else self

    _type_as = torch.Tensor.type_as

    def type_as(self, other):
        out = _type_as(self, other)
        return out.clone() if (out.is_leaf and out.requires_grad) else out

    torch.Tensor.cuda = cuda
    torch.Tensor.type_as = type_as

    dcn = types.ModuleType('DCN')

    def deform_conv_forward(inp, weight, bias, offset, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw, group, dg, step):
        assert group == 1 and dg == 1
        return oracle_dcn.deform_conv3d_forward(inp, offset, weight, bias, (sd, sh, sw), (pd, ph, pw), (dd, dh, dw))

    def deform_conv_backward(inp, weight, bias, offset, go, kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw, group, dg, step):
        return list(oracle_dcn.deform_conv3d_backward(inp, offset, weight, bias, go.contiguous(),
                                                      (sd, sh, sw), (pd, ph, pw), (dd, dh, dw)))

    dcn.deform_conv_forward = deform_conv_forward
    dcn.deform_conv_backward = deform_conv_backward
    sys.modules['DCN'] = dcn


class Obj(object):
    def __init__(self, d):
        for k, v in d.items():
            setattr(self, k, Obj(v) if isinstance(v, dict) else v)


def load_option(model_name='stereodpnet', **model_over):
    data = json.load(open(REF / 'config_' / 'train_faceDP.json'))
    data['model_name'] = model_name
    data['load_model'] = None
    data['model'] = json.load(open(REF / 'src' / 'model' / model_name / 'config.json'))
    data['dataset'] = json.load(open(REF / 'dataloader' / 'FaceDP' / 'config.json'))
    data['model']['metric_type'] = []
    data['model'].update(model_over)
    return Obj(data)


def build_reference(model_name='stereodpnet', **model_over):
    os.chdir(REF)
    for p in (str(REF), str(REF / 'src' / 'module' / 'dcn3d')):
        if p not in sys.path:
            sys.path.insert(0, p)
    opt = load_option(model_name, **model_over)
    ns = run_path(str(Path('src/model') / model_name / 'mainmodel.py'))
    model = ns[model_name.upper()](opt)
    return model, opt


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def checksum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


# ----------------------------------------------------------------------------- fixtures
def gen_e2e(tag, B, H, W, train, mask_mode, stages, prepare=None, compact=False):
    """compact: large fixture (c1's 256x256) -- the inputs are not stored (synthetic_batch(B, H, W, seed=0, mask_mode) regenerates them bit
    for bit) and the predictions are kept as every-other-pixel samples plus fp64 checksums."""
    torch.manual_seed(1)
    model, opt = build_reference()
    fill_by_recipe(model)
    if prepare is not None:
        prepare(model)
    model.train(train)
    batch = synthetic_batch(B, H, W, seed=0, mask_mode=mask_mode)
    cap = OrderedDict()
    hooks = []

    def grab(name, first_only=True):
        def fn(mod, inp, out):
            key = name
            n = 0
            while key in cap:
                n += 1
                key = '%s#%d' % (name, n)
            if first_only and n > 1:
                return
            cap[key] = out
        return fn

    hooks.append(model.feature_extraction.register_forward_hook(grab('fea')))
    hooks.append(model.cost_volume.shifting_layer.register_forward_hook(grab('shift')))
    hooks.append(model.cost_volume.attention_layer.register_forward_hook(grab('attn')))
    hooks.append(model.cost_volume.register_forward_hook(grab('volume')))
    hooks.append(model.aggregation.dres1.register_forward_hook(grab('dres1')))
    hooks.append(model.aggregation.register_forward_hook(grab('agg')))
    hooks.append(model.normal_estimator.deform_conv1.register_forward_hook(grab('dcn1')))
    hooks.append(model.normal_estimator.deform_conv1.register_forward_pre_hook(
        lambda m, inp: cap.__setitem__('anm_volume', inp[0])))
    hooks.append(model.normal_estimator.deform_conv2.register_forward_hook(grab('dcn2')))

    # the ANM level selection (normal_module.py:118-136): sample_with_sort returns the selected levels' disparities = costrange[sorted indices];
    # the bound method is wrapped (out of the reference tree) and the indices recovered from those values
    ne = model.normal_estimator
    if hasattr(ne, 'sample_with_sort'):
        orig_sws = ne.sample_with_sort

        def sws(cost, value):
            res_ = orig_sws(cost, value)
            cap['anm_sdisp'] = res_[1].detach().clone()
            return res_
        ne.sample_with_sort = sws
    res = model(batch)
    for h in hooks:
        h.remove()

    out = {}
    if compact:
        out['batch_args'] = np.array([B, H, W, 0])                       # synthetic_batch(B, H, W, seed=0, mask_mode=...)
        out['mask_mode'] = np.array(mask_mode)
        for k in ('pred_depth', 'pred_normal', 'ref_feature'):
            out[k + '_s'] = f32(res[k][..., ::2, ::2])
            out[k + '_cs'] = checksum(res[k])
    else:
        for k, v in batch.items():
            out['in_' + k] = f32(v)
        out['pred_depth'] = f32(res['pred_depth'])
        out['pred_normal'] = f32(res['pred_normal'])
        out['ref_feature'] = f32(res['ref_feature'])
    if 'anm_sdisp' in cap:
        cr = model.normal_estimator.costrange.detach().reshape(1, -1, 1, 1, 1)                  # [1, L, 1, 1, 1]
        out['anm_idx'] = (cap['anm_sdisp'].unsqueeze(1) - cr).abs().argmin(1).to(torch.uint8).numpy()      # [B, K, h, w]
    out['prob_depth_cs'] = checksum(res['prob_depth'])
    out['prob_depth_s'] = f32(res['prob_depth'][:, :, ::4, ::8, ::8])
    if stages:
        out['fea_ref'] = f32(cap['fea'])          # first call = reference view (right image, flip_lr)
        out['fea_tar'] = f32(cap['fea#1'])
        for j, nm in enumerate(('nearest', 'bilinear', 'phase')):
            out['shift_fwd_' + nm] = f32(cap['shift'][j][..., 0])
            out['shift_bwd_' + nm] = f32(cap['shift#1'][j][..., 0])
        out['attn_fwd'] = f32(cap['attn'])
        out['attn_bwd'] = f32(cap['attn#1'])
        out['volume'] = f32(cap['volume'])
        out['cost0_pre'] = f32(cap['dres1'])      # dres1(cost0) before the residual add
        cost_i, cost = cap['agg']
        out['out3'] = f32(cost[0])
        out['logit3_cs'] = checksum(cost_i[0])
        out['logit3_s'] = f32(cost_i[0][:, ::4, ::4, ::4])
        out['anm_volume'] = f32(cap['anm_volume'])
        out['dcn1_out'] = f32(cap['dcn1'][0])
        out['dcn1_offset'] = f32(cap['dcn1'][1])
        out['dcn2_out_cs'] = checksum(cap['dcn2'][0])
    if train:
        for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
            out[k] = f32(res[k])
        opt_ = model.configure_optimizers()[0][0]
        opt_.zero_grad()
        res['final_loss'].backward()
        names, cs = [], []
        for n, p in model.named_parameters():
            if p.grad is None:
                continue
            names.append(n)
            cs.append(checksum(p.grad))
        out['grad_names'] = np.array(names)
        out['grad_cs'] = np.stack(cs)
        full = ['aggregation.classif3.2.weight', 'cost_volume.attention_layer.mask_convs.0.weight',
                'cost_volume.attention_layer.normalize.weight', 'normal_estimator.deform_conv1.bias',
                'normal_estimator.deform_conv1.conv_offset.bias', 'normal_estimator.n_convs.5.0.weight',
                'feature_extraction.firstconv.0.0.weight', 'feature_extraction.block1.prelu.weight',
                'feature_extraction.fpn.inner_blocks.0.bias', 'aggregation.dres2.conv6.0.weight']
        pd = dict(model.named_parameters())
        for n in full:
            out['grad::' + n] = f32(pd[n].grad)
        opt_.step()
        sd = model.state_dict()
        out['post_names'] = np.array(sorted(sd.keys()))
        out['post_cs'] = np.stack([checksum(sd[k].float()) for k in sorted(sd.keys())])
        for n in ('cost_volume.attention_layer.mask_convs.1.running_mean',
                  'cost_volume.attention_layer.mask_convs.1.running_var',
                  'cost_volume.attention_layer.mask_convs.1.num_batches_tracked',
                  'feature_extraction.firstconv.0.1.running_mean'):
            out['post::' + n] = sd[n].detach().cpu().numpy()
    np.savez_compressed(OUT / ('e2e_%s.npz' % tag), **out)
    print('wrote', tag, {k: getattr(v, 'shape', None) for k, v in list(out.items())[:0]})


def gen_state_keys():
    model, opt = build_reference()
    sd = model.state_dict()
    keys = OrderedDict((k, list(v.shape)) for k, v in sd.items())
    json.dump(keys, open(OUT / 'state_dict_keys.json', 'w'), indent=0)
    print('state_dict keys:', len(keys), 'params:', sum(p.numel() for p in model.parameters()))


def gen_loss():
    model, opt = build_reference()
    lm = model.loss_model
    out = {}
    for mode in ('ones', 'bern'):
        g = torch.Generator().manual_seed(7 if mode == 'ones' else 8)
        B, H, W = 2, 24, 40
        pred_depth = (torch.randn(B, 3, H, W, generator=g) * 2).requires_grad_()
        pred_normal = (torch.rand(B, 1, 3, H, W, generator=g) * 2 - 1).requires_grad_()
        batch = synthetic_batch(B, H, W, seed=3, mask_mode=mode)
        res = lm.forward({'pred_depth': pred_depth, 'pred_normal': pred_normal}, batch)
        res['final_loss'].backward()
        out[mode + '_pred_depth'] = f32(pred_depth)
        out[mode + '_pred_normal'] = f32(pred_normal)
        for k in ('disp', 'normal', 'mask'):
            out[mode + '_' + k] = f32(batch[k])
        for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
            out[mode + '_' + k] = f32(res[k])
        out[mode + '_g_pred_depth'] = f32(pred_depth.grad)
        out[mode + '_g_pred_normal'] = f32(pred_normal.grad)
    # the confidence-weighted smooth-L1 (smoothL1.py:33-36; no loader of the reference emits 'conf'): its own small fixture
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 24, 40
    pred_depth = (torch.randn(B, 3, H, W, generator=g) * 2).requires_grad_()
    pred_normal = (torch.rand(B, 1, 3, H, W, generator=g) * 2 - 1).requires_grad_()
    batch = synthetic_batch(B, H, W, seed=4, mask_mode='bern')
    batch['conf'] = torch.rand(B, H, W, generator=g)
    res = lm.forward({'pred_depth': pred_depth, 'pred_normal': pred_normal}, batch)
    res['final_loss'].backward()
    out['conf_pred_depth'], out['conf_pred_normal'] = f32(pred_depth), f32(pred_normal)
    for k in ('disp', 'normal', 'mask', 'conf'):
        out['conf_' + k] = f32(batch[k])
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        out['conf_' + k] = f32(res[k])
    out['conf_g_pred_depth'], out['conf_g_pred_normal'] = f32(pred_depth.grad), f32(pred_normal.grad)
    np.savez_compressed(OUT / 'loss.npz', **out)
    print('wrote loss')


def gen_psmnet_volume():
    os.chdir(REF)
    sys.path.insert(0, str(REF))
    ns = run_path(str(REF / 'src' / 'model' / 'psmnet' / 'modules.py'))
    out = {}
    g = torch.Generator().manual_seed(11)
    ref = torch.randn(2, 40, 12, 20, generator=g)
    tar = torch.randn(2, 40, 12, 20, generator=g)
    out['ref'] = f32(ref)
    out['tar'] = f32(tar)
    for style in ('psmnet', 'gwcnet'):
        opt = Obj({'model': {'cost_volume': style, 'level': 8, 'group_num': 40}})
        cv = ns['CostVolume'](opt, -4, 12)
        out['vol_' + style] = f32(cv(ref, tar))
    np.savez_compressed(OUT / 'psmnet_volume.npz', **out)
    print('wrote psmnet volume')


if __name__ == '__main__':
    install_shims()
    torch.set_num_threads(8)
    if sys.argv[1:] == ['loss']:                        # only tests/golden/loss.npz
        gen_loss()
        sys.exit(0)
    gen_state_keys()
    gen_loss()
    gen_psmnet_volume()
    gen_e2e('train_32x48_b2', 2, 32, 48, True, 'bern', stages=True)
    gen_e2e('eval_32x48_b2', 2, 32, 48, False, 'ones', stages=False)
    gen_e2e('train_64x96_b1', 1, 64, 96, True, 'ones', stages=False)
    gen_e2e('train_128x128_b2', 2, 128, 128, True, 'bern', stages=False)      # better conditioned BatchNorm: gradient / Adam-step pin
    gen_e2e('train_256x256_b1', 1, 256, 256, True, 'bern', stages=False, compact=True)   # BASELINE configs[0]'s size: full 32-wide tiles
