"""Evaluation metrics against the reference's own functions (golden fixture) and closed forms."""
import os

import numpy as np
import pytest
import torch

from dualpixelface_amd import metrics as M
from dualpixelface_amd.config import load_option
from dualpixelface_amd.selectors import metric_selector


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))


def test_absolute_dp_matches_reference(gold):
    out = M.depth_errors(torch.from_numpy(gold['gt']), torch.from_numpy(gold['pred']), torch.from_numpy(gold['mask']), 1.01)
    np.testing.assert_allclose(out, gold['abs_out'], rtol=2e-5)


def test_normal_dp_matches_reference(gold):
    gn, pn, mask = torch.from_numpy(gold['gn']), torch.from_numpy(gold['pn']), torch.from_numpy(gold['mask']).unsqueeze(1)
    out = [M.normal_error_mean(gn, pn, mask), M.normal_error_rmse(gn, pn, mask)]
    np.testing.assert_allclose(out, gold['normal_out'], rtol=2e-5)


def test_affine_dp_closed_forms():
    g = torch.Generator().manual_seed(3)
    x = torch.rand(40, 50, generator=g)
    w = torch.rand(40, 50, generator=g) + 0.1
    # an exact affine relation is recovered: all three metrics vanish
    y = 2.5 * x - 0.7
    assert M.affine_inv_wmae(x, y, w) < 1e-5 and M.affine_inv_wrmse(x, y, w) < 1e-5
    assert abs(1.0 - M.spearman_rank_correlation(x, y, w)) < 1e-9
    assert abs(1.0 - M.spearman_rank_correlation(x, -y, w)) < 1e-9          # the maximum over both rank directions
    # weighted RMSE against numpy's weighted least squares
    y = 2.5 * x - 0.7 + 0.1 * torch.randn(40, 50, generator=g)
    sw = np.sqrt(w.numpy().reshape(-1).astype(np.float64))
    A = np.stack([x.numpy().reshape(-1), np.ones(x.numel())], 1) * sw[:, None]
    coef = np.linalg.lstsq(A, y.numpy().reshape(-1) * sw, rcond=None)[0]
    res = (x.numpy().reshape(-1) * coef[0] + coef[1] - y.numpy().reshape(-1)) ** 2
    ref = np.sqrt((w.numpy().reshape(-1) * res).sum() / w.numpy().sum())
    assert abs(M.affine_inv_wrmse(x, y, w) - ref) < 1e-6
    # the IRLS fit approaches the L1 optimum: not worse than the L2 fit's weighted MAE
    l2_mae = (w.numpy().reshape(-1) * np.sqrt(res)).sum() / w.numpy().sum()
    assert M.affine_inv_wmae(x, y, w) <= l2_mae + 1e-6
    # unweighted Spearman equals scipy's on tie-free data
    from scipy.stats import spearmanr
    rho = spearmanr(x.numpy().reshape(-1), y.numpy().reshape(-1))[0]
    assert abs(M.spearman_rank_correlation(x, y, torch.ones_like(x)) - abs(rho)) < 1e-3


def test_metric_selector_hook_like_the_reference():
    """metric_selector(option).forward(results, batch) -> {name: row}; rows accumulate; viewer prints the means."""
    opt = load_option()
    sel = metric_selector(opt)
    assert sel.metric_name == ['absolute_dp', 'affine_dp', 'normal_dp']      # src/model/stereodpnet/config.json:3
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 16, 24
    ab = torch.tensor([[32.98, -26996.49]] * B)
    disp = torch.rand(B, H, W, generator=g) * 6 - 2
    depth = M.disp2depth(disp.unsqueeze(1), ab)[:, 0]
    batch = {'abvalue': ab, 'disp': disp, 'depth': depth, 'idepth': M.inverse_depth(depth), 'mask': torch.ones(B, H, W),
             'normal': torch.nn.functional.normalize(torch.randn(B, 3, H, W, generator=g), dim=1)}
    results = {'pred_depth': (disp + 0.01 * torch.randn(B, H, W, generator=g)).unsqueeze(1),
               'pred_normal': batch['normal'].unsqueeze(1) + 0.05 * torch.randn(B, 1, 3, H, W, generator=g)}
    out = sel.forward(results, batch)
    assert set(out) == {'absolute_dp', 'affine_dp', 'normal_dp'}
    assert len(out['absolute_dp']) == 8 and len(out['affine_dp']) == 3 and len(out['normal_dp']) == 2
    assert out['absolute_dp'][0] < 0.05 and out['normal_dp'][0] < 10.0 and out['affine_dp'][2] < 0.01
    perfect = sel.forward({'pred_depth': disp.unsqueeze(1), 'pred_normal': batch['normal'].unsqueeze(1)}, batch, log=False)
    assert perfect['absolute_dp'][0] < 1e-5 and perfect['absolute_dp'][5] == 1.0 and perfect['normal_dp'][0] < 0.1
    assert sel.metric_func[0].index == 1                                    # log=False did not accumulate
    sel.viewer()
    sel.metric_func[0].clear()
    assert sel.metric_func[0].index == 0
