"""ORACLE (test infrastructure only) -- CPU restatement of the reference's PSMNet plugin (BASELINE configs[3]):
/root/reference/src/model/psmnet/{mainmodel.py:30-111, modules.py:14-416}.  The aggregation stack and the disparity regression are
the very classes StereoDPNet uses (same code in both model folders), so they are inherited from oracle/stereodpnet.py.

Pinned by tests/test_oracle_golden.py against tests/golden/psmnet_256x256_b2.npz, produced by importing the reference
(tests/golden/make_golden_psmnet.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/."""
import torch
import torch.nn.functional as F

from .psmnet_volume import psm_volume
from .stereodpnet import Cfg, StereoDPNetOracle


class PSMNetOracle(StereoDPNetOracle):
    def __init__(self, state, cfg=None, training=True, cost_volume='psmnet', group_num=40):
        super(PSMNetOracle, self).__init__(state, cfg or Cfg(lambdas=(1.0,)), training)
        self.style, self.group_num = cost_volume, group_num

    def basic_block(self, x, p, stride, pad, dil, downsample):
        """BasicBlock.forward (psmnet/modules.py:14-34)."""
        out = F.relu(self.convbn2(x, p + '.conv1.0', stride, pad, dil))
        out = self.convbn2(out, p + '.conv2', 1, pad, dil)
        if downsample:
            x = self.bn(F.conv2d(x, self.S[p + '.downsample.0.weight'], None, stride), p + '.downsample.1')
        return out + x

    def layer(self, x, p, cin, planes, blocks, stride, pad, dil):
        """feature_extraction._make_layer (psmnet/modules.py:126-139)."""
        x = self.basic_block(x, p + '.0', stride, pad, dil, stride != 1 or cin != planes)
        for i in range(1, blocks):
            x = self.basic_block(x, '%s.%d' % (p, i), 1, pad, dil, False)
        return x

    def feature_extraction(self, img):
        """feature_extraction.forward (psmnet/modules.py:141-168)."""
        S, p, c = self.S, 'feature_extraction', self.cfg.inplanes
        x = F.relu(self.convbn2(img, p + '.firstconv.0', 2, 1, 1))
        x = F.relu(self.convbn2(x, p + '.firstconv.2', 1, 1, 1))
        x = F.relu(self.convbn2(x, p + '.firstconv.4', 1, 1, 1))
        x = self.layer(x, p + '.layer1', c, c, 3, 1, 1, 1)
        raw = self.layer(x, p + '.layer2', c, 2 * c, c // 2, 2, 1, 1)
        x = self.layer(raw, p + '.layer3', 2 * c, 4 * c, 3, 1, 1, 1)
        skip = self.layer(x, p + '.layer4', 4 * c, 4 * c, 3, 1, 1, 2)
        size = skip.shape[2:]
        branches = []
        for i, k in ((1, 2 * c), (2, c), (3, c // 2), (4, c // 4)):          # AvgPool2d kernels 64 / 32 / 16 / 8
            b = F.avg_pool2d(skip, (k, k), (k, k))
            b = F.relu(self.bn(F.conv2d(b, S['%s.branch%d.1.0.weight' % (p, i)]), '%s.branch%d.1.1' % (p, i)))
            branches.append(F.interpolate(b, size=size, mode='bilinear', align_corners=True))
        feat = torch.cat((raw, skip, branches[3], branches[2], branches[1], branches[0]), 1)
        feat = F.relu(self.convbn2(feat, p + '.lastconv.0', 1, 1, 1))
        return F.conv2d(feat, S[p + '.lastconv.2.weight'])

    def losses(self, pred_depth, batch):
        """loss_selector with loss_type ['smoothL1'], lambdas [1.0] (psmnet/config.json; smoothL1.py:15-49)."""
        mask = batch['mask'] > 0
        n = pred_depth.shape[1]
        wts = [1.0] if n == 1 else list(self.cfg.loss_weight)
        sl1 = sum(wts[i] * F.smooth_l1_loss(pred_depth[:, i][mask], batch['disp'][mask]) for i in range(n))
        return {'smoothL1_loss': sl1, 'abvalue': batch['abvalue'], 'final_loss': self.cfg.lambdas[0] * sl1}

    def forward(self, batch):
        """PSMNET.forward (psmnet/mainmodel.py:67-103)."""
        a, b = ('right', 'left') if self.cfg.flip_lr else ('left', 'right')
        ref = self.feature_extraction(batch[a])
        tar = self.feature_extraction(batch[b])
        vol = psm_volume(ref, tar, self.cfg.costrange, self.group_num if self.style == 'gwcnet' else 0)
        self.taps.update(fea_ref=ref, fea_tar=tar, volume=vol)
        logits, costs = self.aggregation(vol)
        self.taps['out3'] = costs[0]
        preds, probs = self.regression(logits)
        res = {'pred_depth': torch.stack(preds, 1), 'prob_depth': torch.stack(probs, 1), 'ref_feature': ref.max(1)[0]}
        if self.training and 'disp' in batch:
            res.update(self.losses(res['pred_depth'], batch))
        return res
