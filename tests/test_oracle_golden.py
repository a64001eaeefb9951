"""CPU: pin the oracle (oracle/) against the golden vectors captured from the imported reference."""
import numpy as np
import torch

from oracle import recipe_state
from oracle.stereodpnet import StereoDPNetOracle, adam_step
from oracle.psmnet_volume import psm_volume


def _close(a, b, tol, name):
    a = a.detach().double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-6)
    assert err <= tol * scale, '%s: %.3e vs scale %.3e' % (name, err, scale)


def _batch(g, dtype=torch.float32):
    return {k[3:]: torch.from_numpy(g[k]).to(dtype) for k in g.files if k.startswith('in_')}


def test_train_stages_losses_and_counters(golden_dir):
    g = np.load(golden_dir + '/e2e_train_32x48_b2.npz')
    st = recipe_state()
    orc = StereoDPNetOracle(st, training=True)
    res = orc.forward(_batch(g))
    t = orc.taps
    _close(t['fea_ref'], g['fea_ref'], 1e-5, 'fea_ref')
    _close(t['fea_tar'], g['fea_tar'], 1e-5, 'fea_tar')
    for j, nm in enumerate(('nearest', 'bilinear', 'phase')):
        _close(t['shift_fwd'][:, :, j], g['shift_fwd_' + nm], 1e-6, 'shift_fwd_' + nm)
        _close(t['shift_bwd'][:, :, j], g['shift_bwd_' + nm], 1e-6, 'shift_bwd_' + nm)
    _close(t['attn_fwd'], g['attn_fwd'], 1e-5, 'attn_fwd')
    _close(t['attn_bwd'], g['attn_bwd'], 1e-5, 'attn_bwd')
    _close(t['volume'], g['volume'], 1e-5, 'volume')
    _close(t['cost0_pre'], g['cost0_pre'], 1e-4, 'cost0_pre')
    _close(t['out3'], g['out3'], 1e-4, 'out3')
    _close(t['anm_volume'], g['anm_volume'], 1e-4, 'anm_volume')
    _close(t['dcn1_offset'], g['dcn1_offset'], 1e-4, 'dcn1_offset')
    _close(t['dcn1_out'], g['dcn1_out'], 1e-4, 'dcn1_out (derived: reference python + oracle DCN)')
    _close(res['pred_depth'], g['pred_depth'], 1e-4, 'pred_depth')
    _close(res['pred_normal'], g['pred_normal'], 1e-4, 'pred_normal')
    _close(res['ref_feature'], g['ref_feature'], 1e-5, 'ref_feature')
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        _close(res[k], g[k], 1e-5, k)
    # all 8 levels identical (un-keyed grid cache) and 16 BN updates of the shared attention BN (Q6)
    v = t['volume']
    assert all(torch.equal(v[:, :, 0], v[:, :, i]) for i in range(1, 8))
    assert int(st['cost_volume.attention_layer.mask_convs.1.num_batches_tracked']) == 16
    _close(st['cost_volume.attention_layer.mask_convs.1.running_mean'],
           g['post::cost_volume.attention_layer.mask_convs.1.running_mean'], 1e-5, 'attention running_mean')
    _close(st['feature_extraction.firstconv.0.1.running_mean'],
           g['post::feature_extraction.firstconv.0.1.running_mean'], 1e-5, 'firstconv running_mean')
    # gradients: this tiny fixture is BatchNorm-ill-conditioned (fp32 vs fp64 oracle differ by ~1e-2), so compare the
    # reference's per-parameter |grad| sums loosely and a few full gradients
    res['final_loss'].backward()
    names, cs = list(g['grad_names']), g['grad_cs']
    worst = 0.0
    for n, c in zip(names, cs):
        if c[1] < 1e-5:
            continue
        mine = st[str(n)].grad.double().abs().sum().item()
        worst = max(worst, abs(mine - c[1]) / c[1])
    assert worst < 5e-2, worst
    for key in g.files:
        if key.startswith('grad::'):
            ref = torch.from_numpy(g[key]).double()
            if ref.norm() < 1e-6:
                continue
            mine = st[key[6:]].grad.double()
            assert (mine - ref).norm() / ref.norm() < 5e-2, key


def test_eval_and_second_size(golden_dir):
    g = np.load(golden_dir + '/e2e_eval_32x48_b2.npz')
    orc = StereoDPNetOracle(recipe_state(requires_grad=False), training=False)
    with torch.no_grad():
        res = orc.forward(_batch(g))
    assert res['pred_depth'].shape[1] == 1
    _close(res['pred_depth'], g['pred_depth'], 1e-4, 'eval pred_depth')
    _close(res['pred_normal'], g['pred_normal'], 1e-4, 'eval pred_normal')
    g = np.load(golden_dir + '/e2e_train_64x96_b1.npz')
    orc = StereoDPNetOracle(recipe_state(requires_grad=False), training=True)
    with torch.no_grad():
        res = orc.forward(_batch(g))
    _close(res['pred_depth'], g['pred_depth'], 1e-4, '64x96 pred_depth')
    _close(res['final_loss'], g['final_loss'], 1e-5, '64x96 final_loss')


def test_oracle_at_256x256_vs_reference(golden_dir):
    """BASELINE configs[0]'s size: the oracle against the imported reference's 256 x 256 run (tests/golden/make_golden_256.py; inputs are
    regenerated, predictions compared on the stored every-other-pixel samples).  Same op order on the same CPU backend: 1e-4."""
    from dualpixelface_amd.recipe import synthetic_batch
    g = np.load(golden_dir + '/e2e_train_256x256_b1.npz')
    B, H, W, seed = (int(v) for v in g['batch_args'])
    batch = synthetic_batch(B, H, W, seed=seed, mask_mode=str(g['mask_mode']))
    st = recipe_state()
    orc = StereoDPNetOracle(st, training=True)
    res = orc.forward(batch)
    _close(res['pred_depth'][..., ::2, ::2], g['pred_depth_s'], 1e-4, '256 pred_depth')
    # the ANM level selection of the reference itself (normal_module.py:118-136, captured through sample_with_sort): the oracle's top-k
    # restatement picks the same 4 levels at every one of the 64 x 64 quarter-resolution pixels
    assert torch.equal(orc.taps['anm_idx'].long(), torch.from_numpy(g['anm_idx']).long())
    _close(res['pred_normal'][..., ::2, ::2], g['pred_normal_s'], 1e-4, '256 pred_normal')
    _close(res['final_loss'], g['final_loss'], 1e-5, '256 final_loss')
    res['final_loss'].backward()
    for k in g.files:
        if k.startswith('grad::') and np.linalg.norm(g[k]) > 1e-6:
            ref = torch.from_numpy(g[k]).double()
            rel = ((st[k[6:]].grad.double() - ref).norm() / ref.norm()).item()
            assert rel <= 1e-3, (k, rel)


def test_losses_fixture(golden_dir):
    g = np.load(golden_dir + '/loss.npz')
    orc = StereoDPNetOracle({}, training=True)
    for mode in ('ones', 'bern'):
        t = lambda k: torch.from_numpy(g[mode + '_' + k])
        pd, pn = t('pred_depth').requires_grad_(), t('pred_normal').requires_grad_()
        out = orc.losses(pd, pn, {'disp': t('disp'), 'normal': t('normal'), 'mask': t('mask'), 'abvalue': None})
        for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
            _close(out[k], g[mode + '_' + k], 1e-6, k)
        out['final_loss'].backward()
        _close(pd.grad, g[mode + '_g_pred_depth'], 1e-5, 'g pred_depth')
        _close(pn.grad, g[mode + '_g_pred_normal'], 1e-5, 'g pred_normal')


def test_adam_matches_torch_optim():
    torch.manual_seed(0)
    p = torch.randn(257, requires_grad=True)
    q = p.detach().clone()
    opt = torch.optim.Adam([p], lr=1e-4, betas=(0.9, 0.999), eps=1e-5)
    m, v = {'p': torch.zeros(257)}, {'p': torch.zeros(257)}
    for step in (1, 2, 3):
        g = torch.randn(257)
        p.grad = g.clone()
        opt.step()
        adam_step({'p': q}, {'p': g}, m, v, step)
    assert torch.allclose(p.detach(), q, rtol=0, atol=1e-7)


def test_psmnet_volume_fixture(golden_dir):
    g = np.load(golden_dir + '/psmnet_volume.npz')
    ref, tar = torch.from_numpy(g['ref']), torch.from_numpy(g['tar'])
    cr = [i * 0.5 - 1.0 for i in range(8)]
    assert torch.equal(psm_volume(ref, tar, cr, 0), torch.from_numpy(g['vol_psmnet']))
    _close(psm_volume(ref, tar, cr, 40), g['vol_gwcnet'], 1e-6, 'gwcnet')


def test_psmnet_oracle_against_reference(golden_dir):
    """The PSMNet restatement (BASELINE configs[3]) reproduces the imported reference: predictions, loss, gradients, BN buffers."""
    import os
    from oracle import recipe_state
    from oracle.psmnet import PSMNetOracle
    from dualpixelface_amd.recipe import synthetic_batch
    g = np.load(golden_dir + '/psmnet_256x256_b2.npz')
    keys = os.path.join(golden_dir, 'psmnet_state_dict_keys.json')
    batch = synthetic_batch(2, 256, 256, seed=7)
    st = recipe_state(keys_file=keys)
    res = PSMNetOracle(st, training=True).forward(batch)
    _close(res['pred_depth'][:, :, ::2, ::2], g['train_pred_depth_s2'], 2e-4, 'psmnet pred_depth')
    _close(res['ref_feature'], g['train_ref_feature'], 2e-4, 'psmnet ref_feature')
    _close(res['final_loss'], g['final_loss'], 1e-4, 'psmnet loss')
    res['final_loss'].backward()
    for k in g.files:
        if k.startswith('grad::'):
            _close(st[k[6:]].grad, g[k], 5e-3, k)
    _close(st['feature_extraction.branch1.1.1.running_mean'], g['post::feature_extraction.branch1.1.1.running_mean'], 1e-4, 'branch1 running_mean')
    st2 = recipe_state(requires_grad=False, keys_file=keys)
    ev = PSMNetOracle(st2, training=False).forward(batch)
    _close(ev['pred_depth'][:, :, ::2, ::2], g['eval_pred_depth_s2'], 2e-4, 'psmnet eval pred_depth')


def test_fractional_shift_triple_vs_reference(golden_dir):
    """subpixel_shift at fractional deltas (asm.py:59-75,112-125 through the irfft(onesided=False) semantics): the
    oracle's three branches against outputs of fresh reference module instances."""
    g = np.load(golden_dir + '/shift_fractional.npz')
    for ci in range(3):
        fea = torch.from_numpy(g['fea%d' % ci])
        for di, delta in enumerate(g['deltas']):
            for direction, sign in (('forward', 1.0), ('backward', -1.0)):
                near, bil, ph = StereoDPNetOracle.shift_triple(fea, sign * float(delta))
                key = 'c%d_d%d_%s_' % (ci, di, direction)
                _close(near, g[key + 'nearest'], 1e-6, key + 'nearest')
                _close(bil, g[key + 'bilinear'], 1e-6, key + 'bilinear')
                _close(ph, g[key + 'phase'], 2e-6, key + 'phase')


def test_fix_mode_model_vs_reference_with_cleared_grid_cache(golden_dir):
    """Per-level shifts (asm_grid_cache_compat = false) against the reference run with its shift-grid cache cleared before every
    call (tests/golden/make_golden_fixmode.py)."""
    from oracle.stereodpnet import Cfg
    g = np.load(golden_dir + '/e2e_fixmode_train_32x48_b2.npz')
    st = recipe_state()
    orc = StereoDPNetOracle(st, cfg=Cfg(grid_cache_compat=False), training=True)
    res = orc.forward(_batch(g))
    t = orc.taps
    for j, nm in enumerate(('nearest', 'bilinear', 'phase')):
        _close(t['shift_fwd'][:, :, j], g['shift_fwd_' + nm], 2e-6, 'shift_fwd_' + nm)
    _close(t['volume'], g['volume'], 1e-5, 'volume')
    v = t['volume']
    assert not torch.equal(v[:, :, 0], v[:, :, 1])                  # the levels now differ
    _close(res['pred_depth'], g['pred_depth'], 1e-4, 'pred_depth')
    _close(res['pred_normal'], g['pred_normal'], 1e-4, 'pred_normal')
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        _close(res[k], g[k], 1e-5, k)
    _close(st['cost_volume.attention_layer.mask_convs.1.running_mean'],
           g['post::cost_volume.attention_layer.mask_convs.1.running_mean'], 1e-5, 'attention running_mean')
    ge = np.load(golden_dir + '/e2e_fixmode_eval_32x48_b2.npz')
    orc = StereoDPNetOracle(recipe_state(requires_grad=False), cfg=Cfg(grid_cache_compat=False), training=False)
    with torch.no_grad():
        res = orc.forward(_batch(ge))
    _close(res['pred_depth'], ge['pred_depth'], 1e-4, 'eval pred_depth')
    _close(res['pred_normal'], ge['pred_normal'], 1e-4, 'eval pred_normal')


def test_nnet_oracle_against_reference(golden_dir):
    """NNet (src/model/nnet): predictions, normals, losses, gradients and BatchNorm buffers of the oracle against vectors made by
    importing the reference (tests/golden/make_golden_nnet.py)."""
    import os
    from dualpixelface_amd.recipe import synthetic_batch
    from oracle.nnet import NNetOracle
    g = np.load(golden_dir + '/nnet_256x256_b2.npz')
    keys = os.path.join(golden_dir, 'nnet_state_dict_keys.json')
    st = recipe_state(keys_file=keys)
    batch = synthetic_batch(2, 256, 256, seed=11)
    orc = NNetOracle(st, training=True)
    res = orc.forward(batch)
    _close(orc.taps['wc'][:, :3, :, ::4, ::4], g['train_xyz_s'], 1e-5, 'nnet xyz volume')
    _close(orc.taps['costs'][:, :, :, ::2, ::2], g['train_costs_s'], 2e-4, 'nnet costs')
    _close(orc.taps['pool3'][:, :, :, ::4, ::4], g['train_pool3_s'], 2e-4, 'nnet pool3')
    _close(res['pred_depth'][:, :, ::2, ::2], g['train_pred_depth_s2'], 2e-4, 'nnet pred_depth')
    _close(res['pred_normal'][:, :, :, ::2, ::2], g['train_pred_normal_s2'], 2e-4, 'nnet pred_normal')
    _close(res['ref_feature'], g['train_ref_feature'], 2e-4, 'nnet ref_feature')
    for k in ('smoothL1_loss', 'cosine_loss', 'final_loss'):
        _close(res[k], g[k], 1e-4, 'nnet ' + k)
    res['final_loss'].backward()
    for key in g.files:
        if key.startswith('gradcs::'):
            name = key[8:]
            mine = st[name].grad.double()
            ref_abs = g[key][1]
            if ref_abs < 1e-6:
                continue
            assert abs(mine.abs().sum().item() - ref_abs) / ref_abs < 2e-2, (name, mine.abs().sum().item(), ref_abs)
        if key.startswith('grad::'):
            ref = torch.from_numpy(g[key]).double()
            if ref.norm() < 1e-6:
                continue
            mine = st[key[6:]].grad.double()
            assert (mine - ref).norm() / ref.norm() < 2e-2, key
    _close(st['normal_module.pool1.0.1.running_mean'], g['post::normal_module.pool1.0.1.running_mean'], 1e-4, 'pool1 running_mean')
    _close(st['dres2.0.1.running_var'], g['post::dres2.0.1.running_var'], 1e-4, 'dres2 running_var')
    ev = NNetOracle(recipe_state(requires_grad=False, keys_file=keys), training=False)
    with torch.no_grad():
        out = ev.forward(batch)
    _close(out['pred_depth'][:, :, ::2, ::2], g['eval_pred_depth_s2'], 2e-4, 'nnet eval pred_depth')
    _close(out['pred_normal'][:, :, :, ::2, ::2], g['eval_pred_normal_s2'], 2e-4, 'nnet eval pred_normal')


def test_stereonet_oracle_against_reference(golden_dir):
    """StereoNet (src/model/stereonet): logits, both predictions, probabilities, loss, gradients and BatchNorm buffers of the oracle
    against vectors made by importing the reference (tests/golden/make_golden_stereonet.py)."""
    import os
    from dualpixelface_amd.recipe import synthetic_batch
    from oracle.stereonet import StereoNetOracle
    g = np.load(golden_dir + '/stereonet_64x96_b2.npz')
    keys = os.path.join(golden_dir, 'stereonet_state_dict_keys.json')
    st = recipe_state(keys_file=keys)
    batch = synthetic_batch(2, 64, 96, seed=13)
    orc = StereoNetOracle(st, training=True)
    res = orc.forward(batch)
    _close(orc.taps['logits'], g['train_logits'][:, 0], 1e-4, 'stereonet logits')
    _close(res['pred_depth'], g['train_pred_depth'], 1e-4, 'stereonet pred_depth')
    _close(res['prob_depth'], g['train_prob'], 1e-4, 'stereonet prob')
    _close(res['ref_feature'], g['train_ref_feature'], 1e-4, 'stereonet ref_feature')
    _close(res['final_loss'], g['final_loss'], 1e-5, 'stereonet loss')
    res['final_loss'].backward()
    for key in g.files:
        if key.startswith('grad::'):
            ref = torch.from_numpy(g[key]).double()
            if ref.norm() < 1e-6:
                continue
            mine = st[key[6:]].grad.double()
            assert (mine - ref).norm() / ref.norm() < 1e-2, (key, ((mine - ref).norm() / ref.norm()).item())
    assert bool(g['unused_grad_is_none']) and st['feature_extraction.residual_blocks.0.conv2.0.weight'].grad is None
    _close(st['filter.0.0.1.running_mean'], g['post::filter.0.0.1.running_mean'], 1e-4, 'filter.0 running_mean')
    # conv2's BatchNorm is never called: its buffers keep their initial values
    _close(st['feature_extraction.residual_blocks.0.conv2.1.running_var'],
           g['post::feature_extraction.residual_blocks.0.conv2.1.running_var'], 1e-6, 'unused bn')
    ev = StereoNetOracle(recipe_state(requires_grad=False, keys_file=keys), training=False)
    with torch.no_grad():
        out = ev.forward(batch)
    _close(out['pred_depth'], g['eval_pred_depth'], 1e-4, 'stereonet eval pred_depth')


def test_grad_spread_fixture_belongs_to_the_gradient_fixtures_and_pins_the_oracle_in_fp64(golden_dir):
    """tests/golden/grad_spread.npz (make_golden_grad_spread.py: the imported reference at 1 / 2 / 4 / 8 threads and in fp64) is what the GPU
    gradient budgets are derived from.  (i) Its 8-thread run IS the committed e2e fixture (sum g^2 of every gradient equal); (ii) the spreads
    are sane (the head's last layer ~1e-6, a median of a few 1e-3, analytically-zero gradients pure noise); (iii) the ORACLE run in fp64
    reproduces the REFERENCE run in fp64 -- gradients of the well-conditioned 32x48 / batch 2 fixture to 1e-5 (fp32-typed constants inside
    the reference's fp64 run leave ~1e-8 relative input differences, which this network amplifies ~50 x there)."""
    s = np.load(golden_dir + '/grad_spread.npz')
    assert [int(t) for t in s['threads']] == [1, 2, 4, 8]
    for tag in ('train_32x48_b2', 'train_64x96_b1', 'train_128x128_b2'):
        g = np.load(golden_dir + '/e2e_%s.npz' % tag)
        names = [str(n) for n in s[tag + '/names']]
        assert names == [str(n) for n in g['grad_names']]
        np.testing.assert_allclose(s[tag + '/check8'], g['grad_cs'][:, 2], rtol=1e-12, atol=0)
        sp = dict(zip(names, s[tag + '/spread']))
        assert sp['aggregation.classif3.2.weight'] < 5e-6
        assert 5e-4 < float(np.median(s[tag + '/spread'])) < 3e-2
        assert s[tag + '/d64'].shape == (len(names), 4)
    tag = 'train_32x48_b2'
    g = np.load(golden_dir + '/e2e_%s.npz' % tag)
    st = recipe_state(dtype=torch.float64)
    orc = StereoDPNetOracle(st, training=True)
    res = orc.forward({k: (v.double() if v.is_floating_point() else v) for k, v in _batch(g).items()})
    res['final_loss'].backward()
    assert abs(res['final_loss'].item() - float(s[tag + '/loss64'])) <= 1e-7 * abs(float(s[tag + '/loss64']))
    n = 0
    for k in s.files:
        if not k.startswith(tag + '/grad64::'):
            continue
        exact = torch.from_numpy(s[k])
        if exact.norm().item() < 1e-9:
            continue                                                        # analytically zero (a conv bias in front of BatchNorm)
        rel = ((st[k.split('::')[1]].grad - exact).norm() / exact.norm()).item()
        assert rel <= 1e-5, (k, rel)
        n += 1
    assert n >= 9
