"""ORACLE (test infrastructure only) -- CPU restatement of the reference's StereoNet plugin:
/root/reference/src/model/stereonet/{mainmodel.py:30-150, modules.py:10-120}.

Pinned by tests/test_oracle_golden.py against tests/golden/stereonet_64x96_b2.npz, produced by importing the reference
(tests/golden/make_golden_stereonet.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/."""
import torch
import torch.nn.functional as F

from .stereodpnet import Cfg, StereoDPNetOracle

ASTROUS = (1, 2, 4, 8, 1, 1)


class StereoNetOracle(StereoDPNetOracle):
    def __init__(self, state, cfg=None, training=True, k=3):
        super(StereoNetOracle, self).__init__(state, cfg or Cfg(level=2 ** k, lambdas=(1.0,), loss_weight=(1.0, 1.0)), training)
        self.k = k

    def block(self, x, p, dil):
        """BasicBlock.forward (modules.py:19-27): conv2 is constructed but never applied."""
        return x + F.leaky_relu(self.convbn2(x, p + '.conv1.0', 1, 1, dil), 0.2)

    def feature_extraction(self, img):
        """FeatureExtraction.forward (modules.py:53-59)."""
        S, p = self.S, 'feature_extraction'
        x = img
        for i in range(self.k):
            x = F.conv2d(x, S['%s.downsample.%d.weight' % (p, i)], S['%s.downsample.%d.bias' % (p, i)], 2, 2)
        for i in range(6):
            x = self.block(x, '%s.residual_blocks.%d' % (p, i), 1)
        return F.conv2d(x, S[p + '.conv_alone.weight'], S[p + '.conv_alone.bias'], 1, 1)

    def refine(self, low, rgb):
        """EdgeAwareRefinement.forward (modules.py:75-93)."""
        S, p = self.S, 'edge_aware_refinements.0'
        up = F.interpolate(low.unsqueeze(1), size=rgb.shape[-2:], mode='bilinear', align_corners=False)
        if rgb.shape[-1] / low.shape[-1] >= 1.5:
            up = up * 8
        x = F.leaky_relu(self.convbn2(torch.cat([up, rgb], 1), p + '.conv2d_feature.0', 1, 1, 1), 0.2)
        for i, d in enumerate(ASTROUS):
            x = self.block(x, '%s.residual_astrous_blocks.%d' % (p, i), d)
        return F.relu((up + F.conv2d(x, S[p + '.conv2d_out.weight'], S[p + '.conv2d_out.bias'], 1, 1)).squeeze(1))

    def forward(self, batch):
        """STEREONET.forward (mainmodel.py:79-150)."""
        S, cfg = self.S, self.cfg
        a, b = ('right', 'left') if cfg.flip_lr else ('left', 'right')
        ref = self.feature_extraction(batch[a])
        tar = self.feature_extraction(batch[b])
        B, C, h, w = ref.shape
        vol = torch.zeros(B, C, cfg.level, h, w, dtype=ref.dtype)
        parts = []
        for i, disp in enumerate(cfg.costrange):
            d = int(disp)
            lvl = torch.zeros(B, C, h, w, dtype=ref.dtype)
            if d == 0:
                lvl = ref - tar
            elif d > 0:
                lvl = torch.cat([ref[:, :, :-d] - tar[:, :, d:], lvl[:, :, h - d:]], 2)
            else:
                lvl = torch.cat([lvl[:, :, :-d], ref[:, :, -d:] - tar[:, :, :d]], 2)
            parts.append(lvl)
        vol = torch.stack(parts, 2)
        self.taps['volume'] = vol
        x = vol
        for i in range(4):
            x = F.leaky_relu(self.convbn3(x, 'filter.%d.0' % i, 1), 0.2)
        logits = F.conv3d(x, S['conv3d_alone.weight'], S['conv3d_alone.bias'], 1, 1).squeeze(1)
        self.taps['logits'] = logits
        prob = F.softmax(logits, 1)
        L = cfg.level
        disp = torch.tensor([i * ((cfg.maxdisp - cfg.mindisp) / float(L)) + cfg.mindisp for i in range(L)], dtype=torch.float64)
        low = torch.sum(prob * disp.to(prob.dtype).view(1, L, 1, 1), 1)
        right = batch['right']
        refined = self.refine(low, right)
        coarse = F.interpolate((low * (right.shape[-1] / low.shape[-1])).unsqueeze(1), size=right.shape[-2:], mode='bilinear',
                               align_corners=False).squeeze(1)
        res = {'pred_depth': torch.stack([coarse, refined], 1), 'prob_depth': prob.unsqueeze(1), 'ref_feature': ref.max(1)[0]}
        if self.training and 'disp' in batch:
            mask = batch['mask'] > 0
            wts = list(cfg.loss_weight)
            sl1 = sum(wts[i] * F.smooth_l1_loss(res['pred_depth'][:, i][mask], batch['disp'][mask]) for i in range(2))
            res.update({'smoothL1_loss': sl1, 'abvalue': batch['abvalue'], 'final_loss': cfg.lambdas[0] * sl1})
        return res
