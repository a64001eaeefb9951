"""ORACLE (test infrastructure only) -- CPU restatement of the reference's NNet plugin:
/root/reference/src/model/nnet/{mainmodel.py:31-177, modules.py:37-217, normal_module_.py:14-117}.  The ResNet/SPP feature extractor
is PSMNet's with half-pixel resizing of the pyramid branches, so it is inherited from oracle/psmnet.py.

Pinned by tests/test_oracle_golden.py against tests/golden/nnet_256x256_b2.npz, produced by importing the reference
(tests/golden/make_golden_nnet.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/."""
import torch
import torch.nn.functional as F

from .psmnet import PSMNetOracle
from .psmnet_volume import psm_volume
from .stereodpnet import Cfg

REFINE_DIL = (1, 2, 4, 8, 16, 1, 1)          # mainmodel.py:50-58 / normal_module_.py:36-44: seven convtext blocks each


class NNetOracle(PSMNetOracle):
    def __init__(self, state, cfg=None, training=True):
        super(NNetOracle, self).__init__(state, cfg or Cfg(loss_weight=(1.0, 1.0)), training)

    def feature_extraction(self, img):
        """feature_extraction.forward (nnet/modules.py:99-125): as PSMNet's, branches resized with align_corners=False."""
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
        for i, k in ((1, 2 * c), (2, c), (3, c // 2), (4, c // 4)):
            b = F.avg_pool2d(skip, (k, k), (k, k))
            b = F.relu(self.bn(F.conv2d(b, S['%s.branch%d.1.0.weight' % (p, i)]), '%s.branch%d.1.1' % (p, i)))
            branches.append(F.interpolate(b, size=size, mode='bilinear', align_corners=False))
        feat = torch.cat((raw, skip, branches[3], branches[2], branches[1], branches[0]), 1)
        feat = F.relu(self.convbn2(feat, p + '.lastconv.0', 1, 1, 1))
        return F.conv2d(feat, S[p + '.lastconv.2.weight'])

    def convtext_stack(self, f, p):
        """seven convtext blocks (nnet/modules.py:37-42): 3x3 conv, dilation d, padding d, no bias, LeakyReLU(0.1)."""
        for i, d in enumerate(REFINE_DIL):
            f = F.leaky_relu(F.conv2d(f, self.S['%s.%d.0.weight' % (p, i)], None, 1, d, d), 0.1)
        return f

    def residual3(self, x, p):
        r = F.relu(self.convbn3(x, p + '.0', 1))
        return self.convbn3(r, p + '.2', 1) + x

    def normal_module(self, cost_in, batch):
        """NormalModule.forward (normal_module_.py:89-117) with grid_maker_3d (:50-87)."""
        S, p, cfg = self.S, 'normal_module', self.cfg
        B, _, D, h, w = cost_in.shape
        K, abvalue = batch['K'], batch['abvalue']
        disp = torch.tensor(cfg.costrange, dtype=torch.float32).view(1, D, 1, 1).expand(B, D, h, w).to(cost_in.dtype)
        xs = torch.arange(0, w).to(K.dtype)
        ys = torch.arange(0, h).to(K.dtype)
        yg, xg = torch.meshgrid([ys, xs], indexing='ij')
        pix = torch.stack([xg, yg, torch.ones_like(xg)], 0).view(1, 3, h * w).expand(B, -1, -1)
        Kq = K.clone()
        Kq[:, :2, :] = Kq[:, :2, :] / 4.0
        rays = torch.bmm(torch.inverse(Kq), pix).view(B, 3, h, w).to(cost_in.dtype)
        a = abvalue[:, 1].view(B, 1, 1, 1).to(cost_in.dtype)                         # geometry.py:35-40
        b = abvalue[:, 0].view(B, 1, 1, 1).to(cost_in.dtype)
        depth = a / (disp - b)
        depth = torch.where(torch.isnan(depth) | torch.isinf(depth), torch.zeros_like(depth), depth)
        xyz = rays.unsqueeze(2) * depth.unsqueeze(1)
        lo = xyz.reshape(B, -1).min(-1)[0].view(B, 1, 1, 1, 1)
        hi = xyz.reshape(B, -1).max(-1)[0].view(B, 1, 1, 1, 1)
        nxyz = (xyz - lo) / (hi - lo + 1e-6)
        wc = torch.cat((nxyz, cost_in), 1).contiguous()
        self.taps['wc'] = wc
        x = F.relu(self.convbn3(wc, p + '.wc0.0', 1))
        x = F.relu(self.convbn3(x, p + '.wc0.2', 1))
        for n in ('pool1', 'pool2', 'pool3'):                                         # (2,3,3) kernels, stride (2,1,1), padding (0,1,1)
            y = F.conv3d(x, S['%s.%s.0.0.weight' % (p, n)], None, (2, 1, 1), (0, 1, 1))
            x = F.relu(self.bn(y, '%s.%s.0.1' % (p, n)))
        self.taps['pool3'] = x
        nmap = 0
        for i in range(x.shape[2]):
            nmap = nmap + self.convtext_stack(x[:, :, i], p + '.n_convs')
        nmap = F.interpolate(nmap, scale_factor=4, mode='bilinear', align_corners=True)
        return F.normalize(nmap, dim=1)

    def forward(self, batch):
        """NNET.forward (mainmodel.py:112-167)."""
        S = self.S
        a, b = ('right', 'left') if self.cfg.flip_lr else ('left', 'right')
        ref = self.feature_extraction(batch[a])
        tar = self.feature_extraction(batch[b])
        vol = psm_volume(ref, tar, self.cfg.costrange, 0)
        c0 = F.relu(self.convbn3(vol, 'dres0.0', 1))
        c0 = F.relu(self.convbn3(c0, 'dres0.2', 1))
        cost_in0 = c0
        for n in ('dres1', 'dres2', 'dres3', 'dres4'):
            c0 = self.residual3(c0, n)
        cost_in = torch.cat((cost_in0, c0), 1)
        costs = F.conv3d(F.relu(self.convbn3(c0, 'classify.0', 1)), S['classify.2.weight'], None, 1, 1)
        self.taps['costs'] = costs
        refined = []
        for i in range(self.cfg.level):
            costt = costs[:, :, i]
            refined.append(self.convtext_stack(torch.cat([ref, costt], 1), 'convs') + costt)
        costss = torch.stack(refined, 2)
        up = lambda t: F.interpolate(t, scale_factor=4, mode='trilinear', align_corners=False).squeeze(1)
        preds, probs = self.regression([up(costs), up(costss)])
        normal = self.normal_module(cost_in, batch)
        res = {'pred_depth': torch.stack(preds, 1), 'prob_depth': torch.stack(probs, 1), 'pred_normal': normal.unsqueeze(1),
               'ref_feature': ref.max(1)[0]}
        if self.training and 'disp' in batch:
            res.update(StereoLosses.losses(self, res['pred_depth'], res['pred_normal'], batch))
        return res


class StereoLosses(object):
    """the smoothL1 + cosine pair of oracle/stereodpnet.py (loss_selector.py:29-42), PSMNetOracle overrides it with smoothL1 only."""
    from .stereodpnet import StereoDPNetOracle as _Base
    losses = _Base.losses
