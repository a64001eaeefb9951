"""End-to-end GPU parity (``-m gpu``): the HIP StereoDPNet against the golden vectors captured from the imported
reference (tests/golden/e2e_*.npz, recipe weights) and against the CPU oracle's gradients.

Tolerances (fp32, ~100 layers, different summation orders): stage outputs rtol 2e-4 of the tensor scale; predictions
|d disp| <= 2e-3 px (5e-3 px in eval mode, where the recipe's running statistics sharpen the soft-argmin),
|d normal| <= 1e-3; losses rtol 1e-4.  Gradients of this tiny, BatchNorm-ill-conditioned fixture
(12 elements per channel at 1/16 resolution) differ by ~1e-2 between the fp32 and fp64 oracle themselves, so they are
checked to 5e-2 relative L2 against the fp64 oracle on the parameters with a non-negligible gradient.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def build_model(training=True, **model_overrides):
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import fill_by_recipe
    model = STEREODPNET(load_option(**model_overrides))
    fill_by_recipe(model)
    model.to(DEV)
    model.train(training)
    return model


def load_batch(g):
    return {k[3:]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith('in_')}


def close(a, b, tol, name, atol=None):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol if atol is not None else tol * max(b.abs().max().item(), 1e-6)
    assert err <= lim, '%s: max err %.3e > %.3e' % (name, err, lim)


def test_train_forward_stages_and_losses(golden_dir):
    g = np.load(golden_dir + '/e2e_train_32x48_b2.npz')
    model = build_model(True)
    res = model(load_batch(g))
    taps = model.last_taps
    close(taps['fea_ref'], g['fea_ref'], 2e-4, 'fea_ref')
    close(taps['fea_tar'], g['fea_tar'], 2e-4, 'fea_tar')
    close(taps['volume'], g['volume'], 2e-4, 'volume')
    close(taps['out3'], g['out3'], 5e-4, 'out3')
    close(res['pred_depth'], g['pred_depth'], None, 'pred_depth', atol=2e-3)
    close(res['pred_normal'], g['pred_normal'], None, 'pred_normal', atol=1e-3)
    close(res['ref_feature'], g['ref_feature'], 2e-4, 'ref_feature')
    close(res['prob_depth'][:, :, ::4, ::8, ::8], g['prob_depth_s'], None, 'prob_depth', atol=1e-4)
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        close(res[k], g[k], 1e-4, k)
    sd = model.state_dict()
    close(sd['cost_volume.attention_layer.mask_convs.1.running_mean'], g['post::cost_volume.attention_layer.mask_convs.1.running_mean'], 1e-4, 'attn rm')
    close(sd['cost_volume.attention_layer.mask_convs.1.running_var'], g['post::cost_volume.attention_layer.mask_convs.1.running_var'], 1e-4, 'attn rv')
    assert int(sd['cost_volume.attention_layer.mask_convs.1.num_batches_tracked']) == 16
    close(sd['feature_extraction.firstconv.0.1.running_mean'], g['post::feature_extraction.firstconv.0.1.running_mean'], 1e-4, 'first rm')
    assert 'normal_estimator.grid' in sd


def test_eval_forward(golden_dir):
    g = np.load(golden_dir + '/e2e_eval_32x48_b2.npz')
    model = build_model(False)
    with torch.no_grad():
        res = model(load_batch(g))
    assert res['pred_depth'].shape[1] == 1 and 'final_loss' not in res
    close(res['pred_depth'], g['pred_depth'], None, 'pred_depth', atol=5e-3)
    close(res['pred_normal'], g['pred_normal'], None, 'pred_normal', atol=1e-3)


def test_fix_mode_per_level_shifts_vs_reference_with_cleared_grid_cache(golden_dir):
    """asm_grid_cache_compat = false: every cost level uses its own (fractional) shift.  Fixture: the reference run with its
    shift-grid cache cleared before every call (tests/golden/make_golden_fixmode.py); gradients against the fp32 oracle."""
    from oracle import recipe_state
    from oracle.stereodpnet import Cfg, StereoDPNetOracle
    g = np.load(golden_dir + '/e2e_fixmode_train_32x48_b2.npz')
    model = build_model(True, asm_grid_cache_compat=False)
    res = model(load_batch(g))
    taps = model.last_taps
    close(taps['volume'], g['volume'], 2e-4, 'volume')
    assert not torch.equal(taps['volume'][:, :, 0], taps['volume'][:, :, 1])
    close(taps['out3'], g['out3'], 5e-4, 'out3')
    close(res['pred_depth'], g['pred_depth'], None, 'pred_depth', atol=2e-3)
    close(res['pred_normal'], g['pred_normal'], None, 'pred_normal', atol=1e-3)
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        close(res[k], g[k], 1e-4, k)
    sd = model.state_dict()
    close(sd['cost_volume.attention_layer.mask_convs.1.running_mean'], g['post::cost_volume.attention_layer.mask_convs.1.running_mean'], 1e-4, 'attn rm')
    close(sd['cost_volume.attention_layer.mask_convs.1.running_var'], g['post::cost_volume.attention_layer.mask_convs.1.running_var'], 1e-4, 'attn rv')
    assert int(sd['cost_volume.attention_layer.mask_convs.1.num_batches_tracked']) == 16
    # backward through the 16 attention calls and the phase-shift adjoint: gradient of the feature extractor and the attention
    # weights against the fp64 oracle in the same mode
    st = recipe_state(dtype=torch.float64)
    batch64 = {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith('in_')}
    StereoDPNetOracle(st, cfg=Cfg(grid_cache_compat=False), training=True).forward(batch64)['final_loss'].backward()
    model = build_model(True, asm_grid_cache_compat=False)
    model.train_step(load_batch(g))
    pd = dict(model.named_parameters())
    bad, checked = [], 0
    for name in pd:
        if not (name.startswith('cost_volume.') or name.startswith('feature_extraction.lastconv')):
            continue
        ref = st[name].grad
        if ref is None or ref.norm().item() < 1e-6:
            continue
        checked += 1
        rel = (pd[name].grad.detach().cpu().double() - ref).norm().item() / ref.norm().item()
        if rel > 5e-2:
            bad.append((name, rel))
    assert checked >= 6 and not bad, bad
    # eval mode
    ge = np.load(golden_dir + '/e2e_fixmode_eval_32x48_b2.npz')
    model = build_model(False, asm_grid_cache_compat=False)
    with torch.no_grad():
        res = model(load_batch(ge))
    # eval mode sharpens the soft-argmin (recipe running statistics): ONE pixel carries the maximum, and it moves with the summation order of
    # the convolutions -- measured on MI355X (profiles/r05_eval_maxerr_by_matrix_path.txt): fp32 MFMA 3.6e-3 px, six / eight / all nine
    # (= exact products) bf16 partial products 6.6e-3 / 5.5e-3 / 7.2e-3, mean error 1.7e-5 ... 2.0e-5 on every path
    close(res['pred_depth'], ge['pred_depth'], None, 'pred_depth', atol=1.2e-2)
    assert (res['pred_depth'].cpu().double() - torch.from_numpy(ge['pred_depth']).double()).abs().mean().item() <= 5e-5
    close(res['pred_normal'], ge['pred_normal'], None, 'pred_normal', atol=1e-3)


def test_train_64x96(golden_dir):
    g = np.load(golden_dir + '/e2e_train_64x96_b1.npz')
    model = build_model(True)
    res = model(load_batch(g))
    close(res['pred_depth'], g['pred_depth'], None, 'pred_depth', atol=2e-3)
    close(res['pred_normal'], g['pred_normal'], None, 'pred_normal', atol=1e-3)
    close(res['final_loss'], g['final_loss'], 1e-4, 'final_loss')


def test_gradients_and_adam_step_vs_fp64_oracle(golden_dir):
    from oracle import recipe_state
    from oracle.stereodpnet import StereoDPNetOracle
    g = np.load(golden_dir + '/e2e_train_32x48_b2.npz')
    st = recipe_state(dtype=torch.float64)
    batch64 = {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith('in_')}
    orc = StereoDPNetOracle(st, training=True)
    orc.forward(batch64)['final_loss'].backward()
    model = build_model(True)
    p_before = model.flat_parameters().clone()
    res = model.train_step(load_batch(g))
    close(res['final_loss'], g['final_loss'], 1e-4, 'final_loss')
    pd = dict(model.named_parameters())
    bad, checked = [], 0
    for name, off, numel, shape in model._layout:
        ref = st[name].grad
        if ref is None:
            continue
        mine = pd[name].grad.detach().cpu().double()
        rn = ref.norm().item()
        if rn < 1e-6:
            continue
        checked += 1
        rel = (mine - ref).norm().item() / rn
        if rel > (1e-1 if numel == 1 else 5e-2):      # scalar PReLU slopes: a single ill-conditioned sum
            bad.append((name, rel))
    assert checked > 250 and not bad, bad[:10]
    # Adam: first step moves every parameter with a gradient by ~lr (bias-corrected m/sqrt(v) = sign(g))
    delta = (model.flat_parameters() - p_before).abs()
    assert delta.max().item() <= 1.001e-4 and delta.mean().item() > 5e-5


def test_train_step_with_flat_grad_reducer_single_rank(golden_dir):
    """The bucketed all-reduce hooks (world size 1 here: no collective is launched) must see every gradient bucket
    complete during backward, and the fused step must give the same losses as the plain step."""
    from dualpixelface_amd.distributed import make_reducer
    g = np.load(golden_dir + '/e2e_train_32x48_b2.npz')
    model = build_model(True)
    red = make_reducer(model, nbuckets=3)
    assert len(red.buckets) == 3 and sum(red._need) == len(model._layout)
    res = model.train_step(load_batch(g), red)
    close(res['final_loss'], g['final_loss'], 1e-4, 'final_loss')
    assert red._count is None and red._work == []
    res2 = model.train_step(load_batch(g), red)          # second step: hooks re-armed, loss moved
    assert abs(float(res2['final_loss']) - float(res['final_loss'])) > 0
    red.remove()


@pytest.mark.parametrize('shape', [(3, 48, 80), (1, 96, 64)])
def test_other_shapes_against_the_oracle(shape):
    """Ragged sizes (odd batch, H != W, tiles that do not divide the kernels' 8x32 / 2x32 blocks): HIP model vs the CPU
    oracle on the same recipe weights and synthetic batch, train mode, Bernoulli mask."""
    from oracle import recipe_state
    from oracle.stereodpnet import StereoDPNetOracle
    from dualpixelface_amd.recipe import synthetic_batch
    B, H, W = shape
    batch = synthetic_batch(B, H, W, seed=11, mask_mode='bern')
    orc = StereoDPNetOracle(recipe_state(requires_grad=False), training=True)
    with torch.no_grad():
        ref = orc.forward(batch)
    model = build_model(True)
    res = model({k: v.to(DEV) for k, v in batch.items()})
    close(model.last_taps['volume'], orc.taps['volume'], 2e-4, 'volume')
    close(res['pred_depth'], ref['pred_depth'], None, 'pred_depth', atol=3e-3)
    close(res['pred_normal'], ref['pred_normal'], None, 'pred_normal', atol=1e-3)
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        close(res[k], ref[k], 2e-4, k)
    res['final_loss'].backward()          # backward runs at these shapes too (values are covered by the fixture test)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_dcn_compat_module_matches_reference_signature():
    """The `DCN` drop-in (dualpixelface_amd.dcn_compat) called exactly as functions/deform_conv_func.py:27-35,45-56 does."""
    import dualpixelface_amd.dcn_compat as DCN
    from oracle import dcn3d
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 4, 6, 10, generator=g)
    w = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(16, generator=g)
    off = torch.randn(2, 81, 4, 6, 10, generator=g)
    out = DCN.deform_conv_forward(x.to(DEV), w.to(DEV), b.to(DEV), off.to(DEV), 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64)
    close(out, dcn3d.deform_conv3d_forward(x, off, w, b), 1e-4, 'DCN.deform_conv_forward')
    go = torch.randn(out.shape, generator=g)
    grads = DCN.deform_conv_backward(x.to(DEV), w.to(DEV), b.to(DEV), off.to(DEV), go.to(DEV), 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64)
    assert len(grads) == 4
    for a, r in zip(grads, dcn3d.deform_conv3d_backward(x, off, w, b, go)):
        close(a, r, 2e-4, 'DCN.deform_conv_backward')
    with pytest.raises(RuntimeError):
        DCN.deform_conv_forward(x, w.to(DEV), b.to(DEV), off.to(DEV), 3, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 64)   # CPU tensor


def test_validation_step_runs_the_metric_hooks():
    """validation_step -> metric_selector.forward(results, batch) like mainmodel.py:143-148; the metrics of the GPU forward equal
    the metrics of the same prediction evaluated on the host."""
    from dualpixelface_amd import metrics as M
    from dualpixelface_amd.recipe import synthetic_batch
    model = build_model(training=False)
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 32, 48, seed=3).items()}
    with torch.no_grad():
        results = model.validation_step(batch, 0)
    sel = model.metric_model
    assert sel.metric_name == ['absolute_dp', 'affine_dp', 'normal_dp'] and all(f.index == 1 for f in sel.metric_func)
    rows = {n: f.get_value(0) for n, f in zip(sel.metric_name, sel.metric_func)}
    assert all(np.isfinite(v) for r in rows.values() for v in r)
    cpu_res = {k: v.cpu() for k, v in results.items() if torch.is_tensor(v)}
    cpu_batch = {k: v.cpu() for k, v in batch.items()}
    depth = M.disp2depth(cpu_res['pred_depth'], cpu_batch['abvalue'])
    ref = M.depth_errors(cpu_batch['depth'], depth[:, 0], cpu_batch['mask'], 1.01)
    np.testing.assert_allclose(rows['absolute_dp'], ref, rtol=1e-4, atol=1e-6)
    model.validation_epoch_end([results])


@pytest.mark.parametrize('precision', ['bf16', 'bf16-2d'])
def test_bf16_mixed_precision_mode_tracks_fp32(golden_dir, precision):
    """option.precision = 'bf16' / 16 (the reference's PL `precision: 16`: every dense conv with bf16 operands, fp32 accumulation and
    tensors) and 'bf16-2d' (BASELINE config 5 read literally: the 2-D convs only): same graph; outputs within bf16 rounding of the
    fp32 golden run.  8 mantissa bits through ~70 conv + BatchNorm layers of a random-weight network move the soft-argmin by a few
    hundredths of a pixel on average: mean |d disp| <= 0.1 px, max <= 1 px, mean |d normal| <= 0.05, loss within 1 % (2 % with the 3-D
    aggregation in bf16 as well).  The operators themselves are checked to 1e-5 in test_gpu_ops.py."""
    from dualpixelface_amd import load_option, ops
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import fill_by_recipe
    g = np.load(golden_dir + '/e2e_train_32x48_b2.npz')
    opt = load_option()
    opt.precision = precision
    model = STEREODPNET(opt)
    fill_by_recipe(model)
    model.to(DEV).train()
    assert model.bf16_all == (precision == 'bf16') and model.bf16_2d == (precision == 'bf16-2d')
    res = model.forward(load_batch(g))
    assert not ops.CONV_OPERANDS_BF16                     # the precision does not leak out of the forward
    assert torch.isfinite(res['final_loss'])
    d = (res['pred_depth'].detach().cpu() - torch.from_numpy(g['pred_depth'])).abs()
    dn = (res['pred_normal'].detach().cpu() - torch.from_numpy(g['pred_normal'])).abs()
    dl = abs(float(res['final_loss']) - float(g['final_loss'])) / abs(float(g['final_loss']))
    measured = (float(d.mean()), float(d.max()), float(dn.mean()), dl)
    assert float(d.mean()) <= 0.1 and float(d.max()) <= 1.0, measured
    assert float(dn.mean()) <= 0.05, measured
    assert dl <= (2e-2 if precision == 'bf16' else 1e-2), measured
    assert float(d.mean()) > 1e-5, ('bf16 kernels not engaged?', measured)
    res['final_loss'].backward()
    gsum = sum(float(p.grad.abs().sum()) for p in model.parameters() if p.grad is not None)
    assert np.isfinite(gsum) and gsum > 0


def test_psmnet_plugin_against_reference_golden(golden_dir):
    """BASELINE configs[3]: the PSMNet plugin (own feature extractor + integer-shift volume, shared aggregation / head kernels)
    against vectors produced by importing the reference's src/model/psmnet (tests/golden/make_golden_psmnet.py)."""
    import json
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import PSMNET
    from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
    g = np.load(golden_dir + '/psmnet_256x256_b2.npz')
    keys = json.load(open(golden_dir + '/psmnet_state_dict_keys.json'))
    opt = load_option('train_faceDP_psmnet')
    model = PSMNET(opt)
    assert list(model.state_dict().keys()) == list(keys) or set(model.state_dict().keys()) == set(keys)
    fill_by_recipe(model)
    model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 256, 256, seed=7).items()}
    model.flat_gradients(zero=True)
    res = model(batch)
    close(res['pred_depth'][:, :, ::2, ::2], g['train_pred_depth_s2'], None, 'psmnet pred_depth', atol=2e-3)
    close(res['ref_feature'], g['train_ref_feature'], 2e-4, 'psmnet ref_feature')
    close(res['final_loss'], g['final_loss'], 1e-4, 'psmnet loss')
    res['final_loss'].backward()
    pd = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('grad::'):
            ref = torch.from_numpy(g[k]).double()
            rel = ((pd[k[6:]].grad.detach().cpu().double() - ref).norm() / ref.norm()).item()
            # as for StereoDPNet: BatchNorm-ill-conditioned fixture; branch1 normalises 2 values per channel (measured 2.3e-2)
            assert rel <= (1e-1 if 'branch' in k else 5e-2), (k, rel)
    close(model.state_dict()['feature_extraction.branch1.1.1.running_mean'], g['post::feature_extraction.branch1.1.1.running_mean'], 1e-4,
          'branch1 running_mean')
    fill_by_recipe(model)
    model.eval()
    with torch.no_grad():
        ev = model(batch)
    close(ev['pred_depth'][:, :, ::2, ::2], g['eval_pred_depth_s2'], None, 'psmnet eval pred_depth', atol=5e-3)
    # the fused train step (flat arena, Adam) works for this model family too
    model.train()
    r2 = model.train_step(batch, None, lr=1e-4)
    assert torch.isfinite(r2['final_loss'])


def test_stereonet_plugin_against_reference_golden(golden_dir):
    """the StereoNet plugin (5x5 stride-2 feature convs, difference volume, 8-level soft-argmin, full-resolution
    edge-aware refinement with half-pixel bilinear upsampling) against vectors produced by importing the reference's
    src/model/stereonet (tests/golden/make_golden_stereonet.py)."""
    import json
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import STEREONET
    from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
    g = np.load(golden_dir + '/stereonet_64x96_b2.npz')
    keys = json.load(open(golden_dir + '/stereonet_state_dict_keys.json'))
    model = STEREONET(load_option('train_faceDP_stereonet'))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == keys and list(model.state_dict()) == list(keys)
    fill_by_recipe(model)
    model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 64, 96, seed=13).items()}
    model.flat_gradients(zero=True)
    res = model(batch)
    close(model.last_taps['logits'], g['train_logits'], 2e-4, 'stereonet logits')
    close(res['pred_depth'], g['train_pred_depth'], 2e-4, 'stereonet pred_depth')
    close(res['prob_depth'], g['train_prob'], None, 'stereonet prob', atol=1e-5)
    close(res['ref_feature'], g['train_ref_feature'], 2e-4, 'stereonet ref_feature')
    close(res['final_loss'], g['final_loss'], 1e-4, 'stereonet loss')
    res['final_loss'].backward()
    pd = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('grad::'):
            ref = torch.from_numpy(g[k]).double()
            if ref.norm() < 1e-6:
                continue
            mine = pd[k[6:]].grad.detach().cpu().double()
            if k.endswith('conv3d_alone.bias'):
                # a constant added to every level cancels in the softmax: the true gradient is 0 and both sides hold rounding noise
                assert float(mine.abs().max()) <= 1e-3 and float(ref.abs().max()) <= 1e-3, (k, mine, ref)
                continue
            rel = ((mine - ref).norm() / ref.norm()).item()
            assert rel <= 2e-2, (k, rel)
    unused = pd['feature_extraction.residual_blocks.0.conv2.0.weight'].grad
    assert unused is None or float(unused.abs().max()) == 0.0                    # BasicBlock never applies conv2 (modules.py:19-27)
    sd = model.state_dict()
    close(sd['filter.0.0.1.running_mean'], g['post::filter.0.0.1.running_mean'], 1e-4, 'filter.0 rm')
    close(sd['feature_extraction.residual_blocks.0.conv2.1.running_var'],
          g['post::feature_extraction.residual_blocks.0.conv2.1.running_var'], 1e-6, 'unused bn')
    fill_by_recipe(model)
    model.eval()
    with torch.no_grad():
        ev = model(batch)
    close(ev['pred_depth'], g['eval_pred_depth'], 5e-4, 'stereonet eval pred_depth')
    model.train()
    before = model.state_dict()['feature_extraction.residual_blocks.0.conv2.0.weight'].clone()
    r2 = model.train_step(batch, None, lr=1e-4)
    assert torch.isfinite(r2['final_loss'])
    assert torch.equal(before, model.state_dict()['feature_extraction.residual_blocks.0.conv2.0.weight'])   # zero gradient: Adam leaves it


def test_nnet_plugin_against_reference_golden(golden_dir):
    """the NNet plugin (PSMNet-style features with half-pixel pyramid resizing, integer-shift volume, residual 3-D stack,
    per-level 2-D refinement with dilations up to 16, half-pixel trilinear head, plain normal module with (2,3,3) depth-halving convs)
    against vectors produced by importing the reference's src/model/nnet (tests/golden/make_golden_nnet.py)."""
    import json
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import NNET
    from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
    g = np.load(golden_dir + '/nnet_256x256_b2.npz')
    keys = json.load(open(golden_dir + '/nnet_state_dict_keys.json'))
    model = NNET(load_option('train_faceDP_nnet'))
    assert set(model.state_dict().keys()) == set(keys) - {'normal_module.grid'}
    fill_by_recipe(model)
    model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 256, 256, seed=11).items()}
    model.flat_gradients(zero=True)
    res = model(batch)
    assert set(model.state_dict().keys()) == set(keys)                           # the lazily registered grid is there now
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == keys
    close(model.last_taps['costs'][:, :, :, ::2, ::2], g['train_costs_s'], 5e-4, 'nnet costs')
    close(res['pred_depth'][:, :, ::2, ::2], g['train_pred_depth_s2'], None, 'nnet pred_depth', atol=2e-3)
    close(res['pred_normal'][:, :, :, ::2, ::2], g['train_pred_normal_s2'], None, 'nnet pred_normal', atol=1e-3)
    close(res['ref_feature'], g['train_ref_feature'], 2e-4, 'nnet ref_feature')
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        close(res[k], g[k], 1e-4, 'nnet ' + k)
    res['final_loss'].backward()
    pd = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('grad::'):
            ref = torch.from_numpy(g[k]).double()
            if ref.norm() < 1e-6:
                continue
            rel = ((pd[k[6:]].grad.detach().cpu().double() - ref).norm() / ref.norm()).item()
            assert rel <= (1e-1 if 'branch' in k else 5e-2), (k, rel)
        if k.startswith('gradcs::') and g[k][1] > 1e-6:
            mine = pd[k[8:]].grad.detach().cpu().double().abs().sum().item()
            assert abs(mine - g[k][1]) / g[k][1] < 5e-2, (k, mine, g[k][1])
    close(model.state_dict()['normal_module.pool1.0.1.running_mean'], g['post::normal_module.pool1.0.1.running_mean'], 1e-4, 'pool1 rm')
    close(model.state_dict()['dres2.0.1.running_var'], g['post::dres2.0.1.running_var'], 1e-4, 'dres2 rv')
    fill_by_recipe(model)
    model.eval()
    with torch.no_grad():
        ev = model(batch)
    close(ev['pred_depth'][:, :, ::2, ::2], g['eval_pred_depth_s2'], None, 'nnet eval pred_depth', atol=5e-3)
    close(ev['pred_normal'][:, :, :, ::2, ::2], g['eval_pred_normal_s2'], None, 'nnet eval pred_normal', atol=2e-3)
    model.train()
    r2 = model.train_step(batch, None, lr=1e-4)
    assert torch.isfinite(r2['final_loss'])


def test_c2_shape_forward_loss_and_gradients_vs_cpu_oracle(golden_dir):
    """BASELINE configs[1] shape (512x768, one pair): the HIP forward + loss + BACKWARD against the CPU oracle on the same recipe
    weights and synthetic batch -- every kernel (forward, data gradient, weight gradient, deformable conv, normalisation) at its
    production tiling (full 32-wide tiles, 8 disparity planes of 128x192, 16 x 128 x 192 ANM planes), not the 32x48 toy sizes of the
    fixture tests.  ~20-40 s of CPU oracle (forward + backward)."""
    from oracle import recipe_state
    from oracle.stereodpnet import StereoDPNetOracle
    from dualpixelface_amd.recipe import synthetic_batch
    batch = synthetic_batch(1, 512, 768, seed=21, mask_mode='bern')
    st = recipe_state()
    orc = StereoDPNetOracle(st, training=True)
    ref = orc.forward(batch)
    ref['final_loss'].backward()
    ref = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ref.items()}
    model = build_model(True)
    res = model.train_step({k: v.to(DEV) for k, v in batch.items()})          # forward + loss + backward + Adam at this size
    close(model.last_taps['volume'], orc.taps['volume'], 2e-4, 'volume')
    close(res['pred_depth'], ref['pred_depth'], None, 'pred_depth', atol=3e-3)
    # The normal head samples the 4 cost levels nearest to the predicted disparity (normal_module.py:80-138): a DISCONTINUOUS selection.
    # Over 98 304 quarter-resolution pixels a prediction that sits within fp32 rounding of a level boundary can pick the neighbouring level
    # on one side only (tools/c2_flip_probe.py: 2 runs in 30, one pixel each), and that pixel's different cost slices then reach every normal
    # in the head's receptive field and, through the head's gradient, every parameter.  So the comparison is split: (i) the selection itself
    # -- at most 8 pixels may differ; (ii) everything downstream of it with the ORACLE's selection imposed (model.anm_idx_override, a
    # diagnostic hook): then normals, losses and gradients are compared pixel by pixel with no allowance for flips.
    idx_gpu = model.last_anm_idx.cpu().long()
    idx_cpu = orc.taps['anm_idx'].long()
    nflip = int((idx_gpu != idx_cpu).any(1).sum())
    assert nflip <= 8, nflip
    if nflip:
        model = build_model(True)
        model.anm_idx_override = idx_cpu.to(DEV)
        res = model.train_step({k: v.to(DEV) for k, v in batch.items()})
        assert torch.equal(model.last_anm_idx.cpu().long(), idx_cpu)
    err = (res['pred_normal'].detach().cpu().double() - ref['pred_normal'].double()).abs()   # [B, 1, 3, H, W]
    assert float(err.max()) <= 1e-3 and float(err.mean()) <= 2e-5, (float(err.max()), float(err.mean()), nflip)
    close(res['smoothL1_loss'], ref['smoothL1_loss'], 2e-4, 'smoothL1_loss')
    for k in ('cosine_loss', 'final_loss'):
        close(res[k], ref[k], 2e-4, k)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # ---- gradients at production tiling.  With ONE sample per BatchNorm batch this configuration is ill-conditioned in fp32: measured
    # (tools/debug/c2_grad_probe.py) the fp32 CPU oracle sits 1e-1 (median over the 287 parameter gradients) from the fp64 oracle, the HIP
    # path 1e-1 as well, the two fp32 paths 2e-2 from each other.  So the yardstick is the fp64 oracle's gradient (cached by
    # tests/golden/make_golden_c2_fp64.py: 12 full tensors spread over the network + sum g^2 of every gradient): the HIP path may be
    # no further from it than twice the fp32 oracle's own distance, tensor by tensor, and the head's last layer agrees to 1e-5.
    g64 = np.load(golden_dir + '/c2_fp64_grads.npz')
    pd = dict(model.named_parameters())
    for k in g64.files:
        if not k.startswith('grad::'):
            continue
        name = k[6:]
        exact = torch.from_numpy(g64[k]).double()
        if exact.norm().item() < 1e-9:
            continue                                                          # analytically zero (conv bias in front of BatchNorm)
        mine = pd[name].grad.detach().cpu().double()
        e_mine = ((mine - exact).norm() / exact.norm()).item()
        e_ref = ((st[name].grad.double() - exact).norm() / exact.norm()).item()
        assert e_mine <= 2.0 * e_ref + 1e-3, (name, e_mine, e_ref)
        if name.endswith('classif3.2.weight'):
            assert ((mine - st[name].grad.double()).norm() / exact.norm()).item() <= 1e-5, name
    cs = dict(zip((str(n) for n in g64['grad_names']), g64['grad_sumsq']))
    r_mine, r_ref = [], []
    for name, p in pd.items():
        c = cs.get(name, 0.0)
        if p.grad is None or c <= 1e-12 or st[name].grad is None:
            continue
        r_mine.append(abs(float((p.grad.detach().double() ** 2).sum()) - c) / c)
        r_ref.append(abs(float((st[name].grad.double() ** 2).sum()) - c) / c)
    assert len(r_mine) > 250 and float(np.median(r_mine)) <= 2.0 * float(np.median(r_ref)) + 1e-3, (len(r_mine), float(np.median(r_mine)), float(np.median(r_ref)))


def test_train_256x256_vs_reference_fixture(golden_dir):
    """BASELINE configs[0]'s size (one 256 x 256 pair) against what the IMPORTED REFERENCE produced (tests/golden/make_golden_256.py: every
    other pixel of the predictions + fp64 checksums, losses, {sum, sum|.|, sum .^2} of every gradient, 10 full gradients): full 32-wide
    tiles in every conv kernel, 64 x 64 cost planes.  The inputs are regenerated (synthetic_batch is bit-reproducible).  One sample per
    BatchNorm batch makes fp32 gradients noisy (cf. the c2 test): full tensors 8e-2, the head's last layer 1e-5, checksum median 2e-2."""
    from dualpixelface_amd.recipe import synthetic_batch
    g = np.load(golden_dir + '/e2e_train_256x256_b1.npz')
    B, H, W, seed = (int(v) for v in g['batch_args'])
    batch = synthetic_batch(B, H, W, seed=seed, mask_mode=str(g['mask_mode']))
    model = build_model(True)
    res = model.train_step({k: v.to(DEV) for k, v in batch.items()})
    close(res['pred_depth'][..., ::2, ::2], g['pred_depth_s'], None, 'pred_depth', atol=3e-3)
    cs = res['pred_depth'].detach().double()
    assert abs(float(cs.sum()) - g['pred_depth_cs'][0]) <= 3e-4 * cs.numel()
    close(res['ref_feature'][..., ::2, ::2], g['ref_feature_s'], 2e-4, 'ref_feature')
    # the ANM level selection is discontinuous (see the c2 tests): the REFERENCE's own selection is in the fixture -- at most 4 of the 4 096
    # quarter-resolution pixels may differ, and if one does, the step is repeated with the reference's selection imposed, so that everything
    # downstream is compared without any allowance for flips
    idx_ref = torch.from_numpy(g['anm_idx']).long()
    nflip = int((model.last_anm_idx.cpu().long() != idx_ref).any(1).sum())
    assert nflip <= 4, nflip
    if nflip:
        model = build_model(True)
        model.anm_idx_override = idx_ref.to(DEV)
        res = model.train_step({k: v.to(DEV) for k, v in batch.items()})
    err = (res['pred_normal'][..., ::2, ::2].detach().cpu().double() - torch.from_numpy(g['pred_normal_s']).double()).abs()
    assert float(err.max()) <= 1e-3, (float(err.max()), float(err.mean()), nflip)
    close(res['smoothL1_loss'], g['smoothL1_loss'], 2e-4, 'smoothL1_loss')
    for k in ('cosine_loss', 'final_loss'):
        close(res[k], g[k], 2e-4, k)
    pd = dict(model.named_parameters())
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        ref = torch.from_numpy(g[k]).double()
        if ref.norm().item() < 1e-6:
            continue
        mine = pd[k[6:]].grad.detach().cpu().double()
        rel = ((mine - ref).norm() / ref.norm()).item()
        tol = 1e-5 if k.endswith('classif3.2.weight') else 8e-2
        assert rel <= tol, (k, rel, nflip)
    rels = []
    for n, c in zip((str(s) for s in g['grad_names']), g['grad_cs']):
        if n in pd and pd[n].grad is not None and c[2] > 1e-12:
            t = pd[n].grad.detach().double()
            rels.append(abs((t * t).sum().item() - c[2]) / c[2])
    assert len(rels) > 200 and float(np.median(rels)) <= 2e-2, (len(rels), float(np.median(rels)))


# ---- gradient budgets tied to the REFERENCE's own fp32 noise (tests/golden/make_golden_grad_spread.py -> tests/golden/grad_spread.npz).
# For every parameter tensor of the three gradient fixtures the imported reference was run at 1 / 2 / 4 / 8 threads (four fp32 summation
# orders of the same program) and once in fp64.  noise[t] = max(self-spread over the thread pairs, max over the thread counts of the fp32
# run's distance to the fp64 run): how far the reference's own gradient of that tensor moves when nothing but the order of its fp32 sums
# changes.  The HIP path is one more summation order; its distance from the committed (8-thread) fixture may be K_SPREAD x that noise, its
# distance from the reference's fp64 gradient K_FP64 x the reference's own (worst thread count) -- constants stated once, for every tensor
# of every fixture.  A path that loses precision (bf16-rounded operands) sits 30-100 x outside.
K_SPREAD = float(os.environ.get('DPF_TEST_K_SPREAD', 4.0))     # (tests/test_gpu_fallbacks.py runs matrix path 0 at 8: see there)
K_FP64 = 4.0
GRAD_FLOOR = 1e-5          # the head's last layer: noise 1e-6, nothing to amplify it
# Measured on MI355X (round 5, default kernel path): distance / noise of the 10 full tensors: max 2.9 (32x48), 1.8 (64x96), 1.5 (128x128);
# distance to the reference's fp64 gradient / the reference's own: max 3.3, geometric means 1.0 / 0.8.
# Every parameter (289) is also checked through sum g^2.  There the noise estimate itself is coarse for some tensors (at these sizes the
# reference's four thread counts often run the same summation order, so the BatchNorm weights of the 2 x 3-pixel stage show a spread far
# below their true conditioning): at least FRAC_WITHIN of the tensors must be within K_SPREAD x noise and every tensor within K_GROSS x -- a
# missing, doubled or wrongly scaled gradient sits at 100 x or more.
FRAC_WITHIN = 0.97
K_GROSS = 25.0


def ref_noise(golden_dir, tag):
    s = np.load(golden_dir + '/grad_spread.npz')
    names = [str(n) for n in s[tag + '/names']]
    noise = np.maximum(s[tag + '/spread'], s[tag + '/d64'].max(1))
    return s, dict(zip(names, noise)), dict(zip(names, s[tag + '/d64'].max(1))), dict(zip(names, s[tag + '/norm64']))


@pytest.mark.parametrize('tag', ['train_32x48_b2', 'train_64x96_b1', 'train_128x128_b2'])
def test_gradients_and_adam_step_vs_reference_fixture(golden_dir, tag):
    """Gradients and the parameters after ONE Adam step against what the imported reference produced (tests/golden/make_golden.py:
    full gradients of 10 parameters, {sum, sum|.|, sum .^2} of every gradient, and of every state_dict entry after optimizer.step()).
    Through ~100 fp32 conv + BatchNorm layers of this random-weight network the reference's own gradients move by 1e-3 ... 2e-2 when only
    its thread count changes; the budget of each tensor is K_SPREAD x ITS noise in the reference (grad_spread.npz), the head 1e-5."""
    g = np.load(golden_dir + '/e2e_%s.npz' % tag)
    _, noise, _, _ = ref_noise(golden_dir, tag)
    model = build_model(True)
    res = model.train_step(load_batch(g))
    close(res['final_loss'], g['final_loss'], 1e-5, 'final_loss')
    pd = dict(model.named_parameters())
    report = []
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        ref = torch.from_numpy(g[k]).double()
        if ref.norm().item() < 1e-6:
            continue                                                          # analytically zero (conv bias in front of BatchNorm): rounding noise
        mine = pd[k[6:]].grad.detach().cpu().double()
        rel = ((mine - ref).norm() / ref.norm()).item()
        budget = max(K_SPREAD * noise[k[6:]], GRAD_FLOOR)
        report.append((rel / noise[k[6:]], k[6:], rel, noise[k[6:]]))
        assert rel <= budget, (k, rel, budget, noise[k[6:]])
    print(tag, 'distance / reference noise per tensor:', ', '.join('%s %.2f' % (n.split('.')[-3] + '.' + n.split('.')[-1], r) for r, n, _, _ in sorted(report, reverse=True)))
    # every other parameter through its sum of squares (|S_mine - S_ref| / S_ref <= 2 d + d^2 for a relative L2 distance d), same K_SPREAD
    rels, over, ratios = [], [], []
    for n, c in zip((str(s_) for s_ in g['grad_names']), g['grad_cs']):
        if n in pd and pd[n].grad is not None and c[2] > 1e-12 and noise.get(n, 1.0) < 0.25:
            t = pd[n].grad.detach().double()
            r = abs((t * t).sum().item() - c[2]) / c[2]
            rels.append(r)
            d = max(K_SPREAD * noise[n], GRAD_FLOOR)
            ratios.append((0.5 * r / noise[n], n))
            if r > 2 * d + d * d:
                over.append((n, r, d))
    ratios.sort(reverse=True)
    print(tag, 'sum g^2 of every gradient: (|dS| / 2S) / noise -- top', ', '.join('%s %.2f' % (n, r) for r, n in ratios[:8]), '; median %.2f' % ratios[len(ratios) // 2][0])
    assert len(rels) > 250 and len(over) <= (1.0 - FRAC_WITHIN) * len(rels), over[:8]
    assert ratios[0][0] <= K_GROSS and ratios[len(ratios) // 2][0] <= 1.0, ratios[:3]
    assert float(np.median(rels)) <= 5e-3, float(np.median(rels))
    # parameters after one Adam step (lr 1e-4, eps 1e-5: SURVEY a10): per-element mean deviation of every tensor <= 0.2 lr
    sd = model.state_dict()
    checked = 0
    for n, c in zip((str(s_) for s_ in g['post_names']), g['post_cs']):
        if n not in sd or not torch.is_floating_point(sd[n]) or 'running_' in n or n.endswith('.grid'):
            continue
        t = sd[n].detach().double()
        assert abs(t.sum().item() - c[0]) <= 2e-5 * t.numel() + 1e-7 * abs(c[0]), (n, t.sum().item(), c[0])
        assert abs(t.abs().sum().item() - c[1]) <= 2e-5 * t.numel() + 1e-7 * abs(c[1]), n
        checked += 1
    assert checked > 280


@pytest.mark.parametrize('tag', ['train_32x48_b2', 'train_64x96_b1', 'train_128x128_b2'])
def test_gradients_no_worse_than_the_reference_vs_fp64(golden_dir, tag):
    """The yardstick is the REFERENCE run in fp64 (grad_spread.npz: full fp64 gradients of the 10 tensors the fixtures store): the HIP
    gradient's distance to it against the reference's own fp32 distance (the worst of its four thread counts -- they differ by up to 27 x for
    one tensor, so one draw is not a bound).  Per tensor <= K_FP64 x, geometric mean over the tensors <= 2 (measured: 3.3 x for one tensor of the 32x48 fixture,
    geometric means 0.8 - 1.0)."""
    import math
    g = np.load(golden_dir + '/e2e_%s.npz' % tag)
    s, _, d64max, _ = ref_noise(golden_dir, tag)
    model = build_model(True)
    model.train_step(load_batch(g))
    pd = dict(model.named_parameters())
    logs = []
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        exact = torch.from_numpy(s[tag + '/grad64::' + k[6:]]).double()
        if exact.norm().item() < 1e-6:
            continue
        mine = pd[k[6:]].grad.detach().cpu().double()
        e_ref = max(d64max[k[6:]], 1e-7)
        e_mine = ((mine - exact).norm() / exact.norm()).item()
        assert e_mine <= max(K_FP64 * e_ref, GRAD_FLOOR), (k, e_mine, e_ref)
        logs.append(math.log(max(e_mine, 1e-12) / e_ref))
    gm = math.exp(sum(logs) / len(logs))
    print(tag, 'distance to the reference fp64 gradient / the reference fp32 distance: geometric mean %.2f, max %.2f' % (gm, math.exp(max(logs))))
    assert len(logs) >= 8 and gm <= 2.0, gm


def test_c2_batch4_forward_loss_and_gradients_vs_oracle_fixture(golden_dir):
    """BASELINE configs[1] AS STATED -- batch 4 of 512 x 768 pairs -- against the CPU oracle's fp32 run at that size
    (tests/golden/make_golden_c2_b4.py: 3 x 3 minutes and 36 GB of host memory, cached as a fixture; the oracle is pinned to the imported
    reference by the smaller fixtures).  Forward: every 4th pixel of the predictions + checksums, the cost volume's checksums, the ANM level
    selection, the losses.  Backward: 12 full gradients and sum g^2 of every gradient, each within K_SPREAD x the oracle's own fp32 noise in
    that tensor: the largest distance between five fp32 runs of the oracle in different summation orders (8 / 5 / 3 threads, another batch
    order, oneDNN off -- thread counts alone are correlated draws: with them only, the median noise reads 4.9e-3 instead of 1.3e-2).
    The ANM level selection is compared first; if a pixel differs, the step is repeated with the oracle's selection imposed and everything
    is compared without allowance for flips.  Measured on MI355X (round 5, one flipped pixel, selection imposed): distance / noise of the 12
    tensors 0.8 ... 1.2 (one PReLU scalar 0.01) -- the HIP gradient sits AT the oracle's own noise level."""
    from dualpixelface_amd.recipe import synthetic_batch
    g = np.load(golden_dir + '/c2_b4_oracle.npz')
    B, H, W, seed = (int(v) for v in g['batch_args'])
    batch = synthetic_batch(B, H, W, seed=seed, mask_mode=str(g['mask_mode']))
    model = build_model(True)
    res = model.train_step({k: v.to(DEV) for k, v in batch.items()})
    vol = model.last_taps['volume'].detach().double()
    assert abs(float(vol.abs().sum()) - g['volume_cs'][1]) <= 2e-5 * g['volume_cs'][1] and abs(float((vol * vol).sum()) - g['volume_cs'][2]) <= 4e-5 * g['volume_cs'][2]
    close(res['pred_depth'][..., ::4, ::4], g['pred_depth_s'], None, 'pred_depth', atol=3e-3)
    pdd = res['pred_depth'].detach().double()
    assert abs(float(pdd.sum()) - g['pred_depth_cs'][0]) <= 3e-4 * pdd.numel()
    # ANM level selection (discontinuous, see the batch-1 test): (i) at most 8 flipped quarter-resolution pixels per sample; (ii) with the
    # oracle's selection imposed, everything downstream is compared without any allowance for flips
    idx_ref = torch.from_numpy(g['anm_idx']).long()
    nflip = int((model.last_anm_idx.cpu().long() != idx_ref).any(1).sum())
    assert nflip <= 8 * B, nflip
    if nflip:
        model = build_model(True)
        model.anm_idx_override = idx_ref.to(DEV)
        res = model.train_step({k: v.to(DEV) for k, v in batch.items()})
        assert torch.equal(model.last_anm_idx.cpu().long(), idx_ref)
    err = (res['pred_normal'][..., ::4, ::4].detach().cpu().double() - torch.from_numpy(g['pred_normal_s']).double()).abs()   # [B, 1, 3, h, w]
    assert float(err.max()) <= 1e-3, (float(err.max()), nflip)
    close(res['smoothL1_loss'], g['smoothL1_loss'], 2e-4, 'smoothL1_loss')
    for k in ('cosine_loss', 'final_loss'):
        close(res[k], g[k], 2e-4, k)
    # ---- gradients
    names = [str(n) for n in g['grad_names']]
    spread = g['grad_spread']
    noise = {n: float(sp) for n, sp in zip(names, spread)}
    pd = dict(model.named_parameters())
    report = []
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        ref = torch.from_numpy(g[k]).double()
        mine = pd[k[6:]].grad.detach().cpu().double()
        rel = ((mine - ref).norm() / ref.norm()).item()
        budget = max(K_SPREAD * noise[k[6:]], GRAD_FLOOR)
        report.append((rel / noise[k[6:]], k[6:], rel, budget))
    print('c2 batch 4: flipped ANM pixels %d (oracle selection imposed afterwards); distance / oracle noise per tensor:' % nflip, ', '.join('%s %.2f (%.1e)' % (n, r, rel) for r, n, rel, _ in sorted(report, reverse=True)))
    for r, n, rel, budget in report:
        assert rel <= budget, (n, rel, budget, nflip)
    over, rels = [], []
    for n, c in zip(names, g['grad_sumsq']):
        if n in pd and pd[n].grad is not None and c > 1e-12 and noise[n] < 0.25:
            t = pd[n].grad.detach().double()
            r = abs((t * t).sum().item() - c) / c
            rels.append(r)
            d = max(K_SPREAD * noise[n], GRAD_FLOOR)
            if r > 2 * d + d * d:
                over.append((n, r, d))
    print('c2 batch 4: sum g^2 of every gradient: %d tensors, %d beyond K_SPREAD x noise:' % (len(rels), len(over)), over[:8])
    assert len(rels) > 200 and len(over) <= (1.0 - FRAC_WITHIN) * len(rels), over[:8]


def _train_once(g):
    model = build_model(True)
    res = model.train_step(load_batch(g))
    torch.cuda.synchronize()
    out = {'pred_depth': res['pred_depth'].detach().clone(), 'pred_normal': res['pred_normal'].detach().clone(),
           'final_loss': res['final_loss'].detach().clone(), 'anm_idx': model.last_anm_idx.clone()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model, out, grads, state


def test_deterministic_mode_two_runs_are_bitwise_equal(golden_dir):
    """dpf_set_deterministic(1) (VERDICT r4 "missing #1"): two whole train steps from the same weights and batch produce the
    SAME BITS -- predictions, loss, every parameter gradient, every parameter and buffer after the Adam step -- with the default stream
    overlap on (weight gradients on a side stream, feature passes on two streams).  The same step in the default mode is allowed to differ
    (float atomics merge overlapping tiles in arrival order); it must agree with the deterministic one to the run-to-run noise."""
    from dualpixelface_amd import ops
    g = np.load(golden_dir + '/e2e_train_128x128_b2.npz')
    with ops.deterministic_mode():
        assert ops.deterministic()
        _, o1, g1, s1 = _train_once(g)
        _, o2, g2, s2 = _train_once(g)
    assert not ops.deterministic()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    assert set(g1) == set(g2) and len(g1) > 280
    diff = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not diff, diff[:10]
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    _, o3, g3, _ = _train_once(g)                      # default mode: same numbers up to the float-atomic noise
    assert (o1['pred_depth'] - o3['pred_depth']).abs().max().item() <= 2e-3
    assert abs(o1['final_loss'].item() - o3['final_loss'].item()) <= 1e-5 * abs(o3['final_loss'].item())
    tot = torch.cat([g1[n].flatten() for n in sorted(g1)]); tot3 = torch.cat([g3[n].flatten() for n in sorted(g1)])
    assert ((tot - tot3).norm() / tot3.norm()).item() <= 1e-2       # (measured 4.5e-3: the fp32 noise level of this network's gradients)


@pytest.mark.parametrize('tag', ['train_32x48_b2', 'train_128x128_b2'])
def test_deterministic_mode_gradients_vs_reference_fixture(golden_dir, tag):
    """The order-independent kernels compute the same gradients: the reference-derived budgets of
    test_gradients_and_adam_step_vs_reference_fixture hold in deterministic mode too."""
    from dualpixelface_amd import ops
    g = np.load(golden_dir + '/e2e_%s.npz' % tag)
    _, noise, _, _ = ref_noise(golden_dir, tag)
    with ops.deterministic_mode():
        model, out, grads, _ = _train_once(g)
    close(out['final_loss'], g['final_loss'], 1e-5, 'final_loss')
    close(out['pred_depth'], g['pred_depth'], None, 'pred_depth', atol=2e-3)
    for k in g.files:
        if not k.startswith('grad::'):
            continue
        ref = torch.from_numpy(g[k]).double()
        if ref.norm().item() < 1e-6:
            continue
        rel = ((grads[k[6:]].cpu().double() - ref).norm() / ref.norm()).item()
        assert rel <= max(K_SPREAD * noise[k[6:]], GRAD_FLOOR), (k, rel, noise[k[6:]])


def test_train_step_as_one_hip_graph_equals_eager_launches(golden_dir):
    """plugin.train_step captures the whole step (forward, loss, backward on three streams, gradient gather, fused Adam) into ONE HIP graph
    on the third call of a batch shape and replays it afterwards.  In deterministic mode eager steps are reproducible, so five graph-mode
    steps (2 eager + capture + 2 replays... the capture itself executes nothing) must leave EXACTLY the parameters, Adam moments, BatchNorm
    buffers and call counters that five eager steps leave; the learning rate changes on the way (it travels through device memory)."""
    from dualpixelface_amd import ops
    g = np.load(golden_dir + '/e2e_train_128x128_b2.npz')
    batch = load_batch(g)
    lrs = [1e-4, 1e-4, 1e-4, 5e-5, 2e-5]
    outs = []
    with ops.deterministic_mode():
        for graph in (False, True):
            model = build_model(True)
            model.option.step_graph = graph
            losses = []
            for lr in lrs:
                res = model.train_step({k: v.clone() for k, v in batch.items()}, lr=lr)
                losses.append(float(res['final_loss']))
            torch.cuda.synchronize()
            live = getattr(model, '_graph_state', None) is not None and model._graph_state.get('graph') is not None
            assert live == graph, (graph, getattr(model, '_graph_state', None) and model._graph_state.get('failed'))
            outs.append((losses, model.flat_parameters().clone(), model._adam['m'].clone(), model._adam['v'].clone(), model._adam['step'],
                         {k: v.clone() for k, v in model.state_dict().items()}))
    (l0, p0, m0, v0, s0, sd0), (l1, p1, m1, v1, s1, sd1) = outs
    assert s0 == s1 == len(lrs)
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    assert int(sd1['feature_extraction.firstconv.0.1.num_batches_tracked']) == 2 * len(lrs)      # two feature passes per step


def test_graph_replay_is_ordered_with_the_callers_stream(golden_dir):
    """A replayed step and whatever the caller enqueues next on ITS stream are ordered: a snapshot of the parameters taken right behind
    train_step() -- no host synchronisation in between, the GPU queue several steps deep -- holds the finished step.  (The replay runs on
    the step's own stream between explicit event waits; launched into the caller's legacy default stream, later kernels were observed to
    start before the graph had finished.)  Deterministic mode, so the run with a device synchronisation after every step is the oracle."""
    from dualpixelface_amd import ops
    g = np.load(golden_dir + '/e2e_train_128x128_b2.npz')
    batch = load_batch(g)
    snaps = []
    with ops.deterministic_mode():
        for sync in (True, False):
            model = build_model(True)
            model.option.step_graph = True
            for _ in range(4):                                 # eager calls (the first one creates the Adam state), then the capture
                model.train_step({k: v.clone() for k, v in batch.items()})
            torch.cuda.synchronize()
            assert model._graph_state.get('graph') is not None
            got = []
            for _ in range(6):
                model.train_step(batch)
                got.append(model.flat_parameters().clone())    # on the caller's stream, right behind the replay
                if sync:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            snaps.append(got)
    for i, (a, b) in enumerate(zip(*snaps)):
        assert torch.equal(a, b), i
    assert not torch.equal(snaps[0][0], snaps[0][5])


@pytest.mark.parametrize('bn_cat', ['0', '1'])
def test_side_stream_weight_gradients_match_in_line(golden_dir, monkeypatch, bn_cat):
    """DPF_WGRAD_ASYNC: the weight-gradient launches move to a side stream; the gradients that reach Adam must be the same ones.
    bn_cat = '1' also turns on DPF_CONV_BN_CAT, whose backward launches its weight gradients on the MAIN stream while the
    registry sends the others to the side stream: the two share no scratch slab (ops.scratch is keyed by stream)."""
    from dualpixelface_amd import ops
    monkeypatch.setenv('DPF_CONV_BN_CAT', bn_cat)
    g = np.load(golden_dir + '/e2e_train_64x96_b1.npz')
    grads = []
    for flag in (False, True):
        monkeypatch.setattr(ops, 'WGRAD_ASYNC', flag)
        deferred = []
        real = ops.wgrad_async_finish

        def spy():
            out = real()
            deferred.append(len(out))
            return out
        monkeypatch.setattr(ops, 'wgrad_async_finish', spy)
        model = build_model(True)
        model.train_step(load_batch(g))
        torch.cuda.synchronize()
        grads.append(model.flat_gradients(zero=False).clone())
        monkeypatch.setattr(ops, 'wgrad_async_finish', real)
        assert (deferred[0] > 100) == flag, deferred
    ref, got = grads
    rel = (got - ref).norm().item() / ref.norm().item()
    assert rel < 1e-5 and torch.isfinite(got).all(), rel
    # per parameter too: a gradient that never arrived (or arrived twice) would hide in the global norm of 3.7 M values
    model_layout = model._layout
    for name, off, numel, shape in model_layout:
        a, b = ref[off:off + numel], got[off:off + numel]
        n = a.norm().item()
        if n > 1e-6:
            assert (a - b).norm().item() / n < 1e-3, name


@pytest.mark.parametrize('family', ['psmnet', 'nnet'])
def test_other_plugins_honour_bf16_operand_precision(golden_dir, family):
    """option.precision = 'bf16' on the PSMNet / NNet plugins (StereoNet shares their code path; bench.py --model stereonet --precision bf16): same graph, the dense convs round their operands to bf16.  The
    loss stays within 2 % of the plugin's own fp32 golden loss (a different but nearby number proves the kernels engaged), the
    gradients are finite, and the precision does not leak out of the forward."""
    from dualpixelface_amd import load_option, ops
    from dualpixelface_amd.plugin import NNET, PSMNET
    from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
    cls, cfg, gold, bs, hw, seed = {
        'psmnet': (PSMNET, 'train_faceDP_psmnet', 'psmnet_256x256_b2.npz', 2, (256, 256), 7),
        'nnet': (NNET, 'train_faceDP_nnet', 'nnet_256x256_b2.npz', 2, (256, 256), 11),
    }[family]
    g = np.load(golden_dir + '/' + gold)
    opt = load_option(cfg)
    opt.precision = 'bf16'
    model = cls(opt)
    assert model.bf16_all
    fill_by_recipe(model)
    model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in synthetic_batch(bs, hw[0], hw[1], seed=seed).items()}
    model.flat_gradients(zero=True)
    res = model(batch)
    assert not ops.CONV_OPERANDS_BF16
    ref = float(g['final_loss'])
    got = float(res['final_loss'])
    assert np.isfinite(got) and abs(got - ref) <= 2e-2 * abs(ref), (got, ref)
    assert abs(got - ref) > 1e-7 * abs(ref), ('bf16 kernels not engaged?', got, ref)
    res['final_loss'].backward()
    gsum = sum(float(p.grad.abs().sum()) for p in model.parameters() if p.grad is not None)
    assert np.isfinite(gsum) and gsum > 0


def test_graph_replays_and_eager_steps_alternate():
    """A train step that replays as a HIP graph, followed closely by the same model launching kernels one by one (an eager step, a
    validation forward), 200 times (VERDICT r5 item 7 / ADVICE r5).  Everything the model launches eagerly runs on the replays' own stream
    (plugin._behind_replays: the cross-stream event behind a graph launch is not relied upon -- the host-side wait of round 5 is gone), so
    the alternation must (i) stay finite, (ii) train exactly like a twin that takes the same 2 x 20 steps without any graph -- in
    deterministic mode (dpf_set_deterministic: the same bits from the same step) the parameters of the two must agree to 1e-6 after 44
    Adam steps, (iii) leave the validation forward usable."""
    from dualpixelface_amd import load_option, ops
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import fill_by_recipe, synthetic_batch
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 128, 128, seed=2).items()}

    def make(graph):
        opt = load_option()
        opt.step_graph = graph
        m = STEREODPNET(opt)
        fill_by_recipe(m)
        return m.to(DEV)

    with ops.deterministic_mode():
        a, b = make(True), make(False)
        _alternate(a, b, batch)


def _alternate(a, b, batch):
    for _ in range(4):                                   # a: eager warm-ups (the first creates the Adam state), then capture + first replay; b: eager steps
        a.train_step(batch)
        b.train_step(batch)
    assert a._graph_state.get('graph') is not None
    for i in range(20):
        ra = a.train_step(batch)                         # replay
        ea = a._eager_step(batch, None, None)            # eager, right behind it
        b.train_step(batch)
        eb = b.train_step(batch)
    torch.cuda.synchronize()
    la, lb = float(ea['final_loss'].detach()), float(eb['final_loss'].detach())
    assert np.isfinite(la) and abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    pa, pb = a.flat_parameters(), b.flat_parameters()
    assert torch.isfinite(pa).all() and (pa - pb).abs().max().item() <= 1e-6, (pa - pb).abs().max().item()
    for i in range(180):
        ra = a.train_step(batch)
        if i % 3 == 0:
            ea = a._eager_step(batch, None, None)
        elif i % 3 == 1:
            a.eval()
            with torch.no_grad():
                va = a.forward(batch)
            a.train()
    torch.cuda.synchronize()
    assert np.isfinite(float(ra['final_loss'])) and torch.isfinite(a.flat_parameters()).all() and torch.isfinite(va['pred_depth']).all()


def test_c5_bf16_whole_step_full_size():
    """BASELINE configs[4] as a test: the bf16 mixed-precision mode (option.precision = 'bf16': the dense convolutions round
    their operands to bf16 while staging them -- PL's `precision: 16` for nn.Conv2d / nn.Conv3d, reference hook main.py:53 -- fp32 accumulation,
    tensors, cost-volume and normalisation arithmetic) on one GPU's share of that configuration, 8 x 1024 x 1536 pairs, one whole train
    step against the fp32 step from the same weights and batch: loss within 2e-3 relative (measured 2.1e-4), gradient arena cosine >= 0.99
    (measured 0.9963 = 8.6 % relative L2: 8-bit operands through ~100 conv + BatchNorm layers of a random-weight network, whose fp32 gradients
    already move by up to 2 % when only the reference's thread count changes -- grad_spread.npz), everything finite, and the two runs really
    differ (the bf16 kernels engaged)."""
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import synthetic_batch
    batch = {k: v.to(DEV) for k, v in synthetic_batch(8, 1024, 1536, seed=1).items()}
    torch.manual_seed(5)
    base = STEREODPNET(load_option()).to(DEV)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    del base
    out = {}
    for prec in ('f32', 'bf16'):
        opt = load_option()
        opt.precision = prec
        model = STEREODPNET(opt).to(DEV)
        assert model.bf16_all == (prec == 'bf16')
        model.load_state_dict(sd, strict=True)
        res = model.train_step(batch)
        torch.cuda.synchronize()
        out[prec] = (float(res['final_loss'].detach()), model.flat_gradients(zero=False).clone())
        del res, model
        torch.cuda.empty_cache()
    (l32, g32), (l16, g16) = out['f32'], out['bf16']
    assert np.isfinite(l32) and np.isfinite(l16) and torch.isfinite(g32).all() and torch.isfinite(g16).all()
    cos = (torch.dot(g32.double(), g16.double()) / (g32.double().norm() * g16.double().norm())).item()
    print('c5 step at 8 x 1024 x 1536: loss fp32 %.6f bf16 %.6f (rel %.2e), gradient cosine %.6f, |dg|/|g| %.2e'
          % (l32, l16, abs(l16 - l32) / abs(l32), cos, ((g16 - g32).norm() / g32.norm()).item()))
    assert abs(l16 - l32) <= 2e-3 * abs(l32), (l16, l32)
    assert abs(l16 - l32) > 1e-7 * abs(l32), ('bf16 kernels not engaged?', l16, l32)
    assert cos >= 0.99, cos


def test_headline_config_whole_train_step(monkeypatch):
    """BASELINE's headline configuration as a TEST, not only as a bench line: one whole train step (forward + loss + backward + Adam) of
    StereoDPNet on 4 x 1024 x 1536 synthetic pairs.  (i) everything finite; (ii) the default step (weight gradients on a side stream, the
    two feature passes on two streams) and the one-stream step start from the same weights and must agree -- loss 1e-5, disparity 2e-3 px,
    gradient arena 2e-3 relative L2 and every parameter's gradient 5e-2 (the run-to-run noise of the float atomics, DESIGN section 2, is the
    only difference between the two schedules); (iii) state_dict -> load_state_dict(strict) -> state_dict is the identity;
    (iv) CROSS-PATH: the same step with every fp32 product on v_mfma_f32_32x32x2_f32 (dpf_set_f32_matrix_path(0), exact
    fp32 per element) -- the default path (f16 components with the range guards) is tied to it at the headline size, where tiles are full and
    the activations have their real dynamic range: loss 1e-5, disparity 2e-3 px, gradient arena relative L2 within 2 x the distance between
    the two schedules of the default path (+ 1e-4, the run-to-run floor of two one-schedule runs), every parameter 5e-2."""
    from dualpixelface_amd import load_option, ops
    from dualpixelface_amd._lib import lib
    import dualpixelface_amd.stereodpnet as sdn
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import synthetic_batch
    batch = {k: v.to(DEV) for k, v in synthetic_batch(4, 1024, 1536, seed=0).items()}
    torch.manual_seed(3)
    base = STEREODPNET(load_option()).to(DEV)             # the reference's initialisation scheme (what bench.py times)
    sd = {k: v.clone() for k, v in base.state_dict().items()}
    runs = []
    prev_path = lib().cdll.dpf_get_f32_matrix_path()
    assert prev_path == 2                                  # the default the bench line is timed on
    try:
        for two_streams, path in ((True, 2), (False, 2), (False, 0)):
            monkeypatch.setattr(ops, 'WGRAD_ASYNC', two_streams)
            monkeypatch.setattr(sdn, 'FEATURES_TWO_STREAMS', two_streams)
            lib().call('dpf_set_f32_matrix_path', path)
            model = STEREODPNET(load_option()).to(DEV)
            model.load_state_dict(sd, strict=True)
            res = model.train_step(batch)
            torch.cuda.synchronize()
            runs.append((float(res['final_loss']), res['pred_depth'].detach().clone(), model.flat_gradients(zero=False).clone(),
                         model.flat_parameters().clone(), model._layout))
            del res
    finally:
        lib().call('dpf_set_f32_matrix_path', prev_path)
    (l2, d2, g2, p2, layout), (l1, d1, g1, p1, _), (l0, d0, g0, p0, _) = runs
    # (iv) the default path against exact fp32 products, same schedule
    for t in (d0, g0, p0):
        assert torch.isfinite(t).all()
    sched = ((g1 - g2).norm() / g1.norm()).item()
    cross = ((g1 - g0).norm() / g0.norm()).item()
    worst0 = []
    for name, off, numel, _ in layout:
        a, b = g0[off:off + numel], g1[off:off + numel]
        if a.norm().item() > 1e-3 * g0.norm().item():
            worst0.append((((a - b).norm() / a.norm()).item(), name))
    worst0.sort(reverse=True)
    print('headline step, f16 components vs fp32 matrix instruction: loss %.6f / %.6f, disparity %.2e px, |dg|/|g| %.2e (two schedules of the default: %.2e), worst parameters %s'
          % (l1, l0, (d1 - d0).abs().max().item(), cross, sched, worst0[:3]))
    assert abs(l1 - l0) <= 1e-5 * abs(l0), (l1, l0)
    assert (d1 - d0).abs().max().item() <= 2e-3
    assert cross <= 2 * sched + 1e-4, (cross, sched)
    assert worst0[0][0] <= 5e-2, worst0[:5]
    del g0, d0, p0
    for t in (d2, g2, p2, d1, g1, p1):
        assert torch.isfinite(t).all()
    assert np.isfinite(l1) and np.isfinite(l2) and abs(l1 - l2) <= 1e-5 * abs(l1), (l1, l2)
    assert (d1 - d2).abs().max().item() <= 2e-3
    assert ((g1 - g2).norm() / g1.norm()).item() <= 2e-3
    worst = []
    gn = g1.norm().item()
    for name, off, numel, _ in layout:
        a, b = g1[off:off + numel], g2[off:off + numel]
        if a.norm().item() > 1e-3 * gn:                   # a parameter whose gradient carries a visible share of the step
            worst.append((((a - b).norm() / a.norm()).item(), name, a.norm().item() / gn))
    worst.sort(reverse=True)
    print('headline step: loss %.6f / %.6f, |dg|/|g| %.2e, worst parameters %s' % (l2, l1, ((g1 - g2).norm() / g1.norm()).item(), worst[:5]))
    assert len(worst) > 50 and worst[0][0] <= 5e-2, worst[:5]
    assert (p1 - p2).abs().max().item() <= 2.1e-4         # one Adam step: at most lr = 1e-4 either way
    # (iii) state_dict round trip on the stepped model
    sd1 = model.state_dict()
    twin = STEREODPNET(load_option()).to(DEV)
    twin.load_state_dict(sd1, strict=True)
    sd2 = twin.state_dict()
    assert list(sd1.keys()) == list(sd2.keys())
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
