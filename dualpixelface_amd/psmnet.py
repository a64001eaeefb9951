"""PSMNet behind the same plugin surface (BASELINE configs[3] "cross-model plugin check"):
the reference's src/model/psmnet/{mainmodel.py:30-111, modules.py:14-416} on the HIP operator layer.

Only the feature extractor (ResNet-style BasicBlocks + SPP branches, psmnet/modules.py:60-168) and the integer-shift cost
volume (:215-275) are specific to this model; the 3-D hourglass aggregation and the soft-argmin head are the classes StereoDPNet
uses and are inherited from ``StereoDPNetCore``.  Same flat parameter arena, fused Adam and reducer as the flagship model.
"""
from . import ops
from .ops import ACT_NONE, ACT_RELU
from .stereodpnet import StereoDPNetCore, _Spec


def build_psmnet_spec(opt):
    m = opt.model
    c = m.inplanes
    s = _Spec()
    fe = 'feature_extraction'
    s.convbn2(fe + '.firstconv.0', m.input_channel, c)
    s.convbn2(fe + '.firstconv.2', c, c)
    s.convbn2(fe + '.firstconv.4', c, c)

    def layer(p, cin, planes, blocks, stride):
        for i in range(blocks):
            ci = cin if i == 0 else planes
            s.convbn2('%s.%d.conv1.0' % (p, i), ci, planes)
            s.convbn2('%s.%d.conv2' % (p, i), planes, planes)
            if i == 0 and (stride != 1 or cin != planes):
                s.conv('%s.0.downsample.0' % p, planes, cin, (1, 1))
                s.bn('%s.0.downsample.1' % p, planes)

    layer(fe + '.layer1', c, c, 3, 1)                      # psmnet/modules.py:79-82
    layer(fe + '.layer2', c, 2 * c, c // 2, 2)
    layer(fe + '.layer3', 2 * c, 4 * c, 3, 1)
    layer(fe + '.layer4', 4 * c, 4 * c, 3, 1)
    for i in range(1, 5):                                   # SPP branches :84-102
        s.conv('%s.branch%d.1.0' % (fe, i), c, 4 * c, (1, 1))
        s.bn('%s.branch%d.1.1' % (fe, i), c)
    s.convbn2(fe + '.lastconv.0', 10 * c, 4 * c)
    s.conv(fe + '.lastconv.2', c, 4 * c, (1, 1))
    ag = 'aggregation'
    first = 2 * c + (int(m.group_num) if m.cost_volume == 'gwcnet' else 0)
    s.convbn3(ag + '.dres0.0', first, c)
    s.convbn3(ag + '.dres0.2', c, c)
    s.convbn3(ag + '.dres1.0', c, c)
    s.convbn3(ag + '.dres1.2', c, c)
    for n in ('dres2', 'dres3', 'dres4'):
        s.hourglass(ag + '.' + n, c)
    for n in ('classif1', 'classif2', 'classif3'):
        s.convbn3(ag + '.%s.0' % n, c, c)
        s.conv(ag + '.%s.2' % n, 1, c, (3, 3, 3))
    return s


class PSMNetCore(StereoDPNetCore):
    spp_align_corners = True        # psmnet/modules.py:150-163 resizes the pooled branches with align_corners=True

    @staticmethod
    def _spec(option):
        return build_psmnet_spec(option)

    def _basic_block(self, x, p, stride, pad, dil, downsample):
        """BasicBlock.forward (psmnet/modules.py:14-34): conv-bn-relu, conv-bn, + (downsampled) input."""
        P = self._P
        out = self._convbn2(x, p + '.conv1.0', stride, pad, dil, ACT_RELU)
        if downsample:
            x = self._bn(self._conv2d(x, P[p + '.downsample.0.weight'], None, stride), p + '.downsample.1')
        return self._convbn2(out, p + '.conv2', 1, pad, dil, ACT_NONE, None, x)

    def _layer(self, x, p, cin, planes, blocks, stride, pad, dil):
        x = self._basic_block(x, p + '.0', stride, pad, dil, stride != 1 or cin != planes)
        for i in range(1, blocks):
            x = self._basic_block(x, '%s.%d' % (p, i), 1, pad, dil, False)
        return x

    def _features(self, img):
        """feature_extraction.forward (psmnet/modules.py:141-168)."""
        P, p, c = self._P, 'feature_extraction', self.option.model.inplanes
        x = self._convbn2(img, p + '.firstconv.0', 2, 1, 1, ACT_RELU)
        x = self._convbn2(x, p + '.firstconv.2', act=ACT_RELU)
        x = self._convbn2(x, p + '.firstconv.4', act=ACT_RELU)
        x = self._layer(x, p + '.layer1', c, c, 3, 1, 1, 1)
        raw = self._layer(x, p + '.layer2', c, 2 * c, c // 2, 2, 1, 1)
        x = self._layer(raw, p + '.layer3', 2 * c, 4 * c, 3, 1, 1, 1)
        skip = self._layer(x, p + '.layer4', 4 * c, 4 * c, 3, 1, 1, 2)
        h, w = skip.shape[2], skip.shape[3]
        branches = []
        for i, k in ((1, 2 * c), (2, c), (3, c // 2), (4, c // 4)):
            b = ops.avg_pool2d(skip, k)
            b = self._bn(self._conv2d(b, P['%s.branch%d.1.0.weight' % (p, i)]), '%s.branch%d.1.1' % (p, i), ACT_RELU)
            branches.append(ops.resize_bilinear(b, h, w, self.spp_align_corners))
        feat = ops.concat_channels([raw, skip, branches[3], branches[2], branches[1], branches[0]])
        feat = self._convbn2(feat, p + '.lastconv.0', act=ACT_RELU)
        return self._conv2d(feat, P[p + '.lastconv.2.weight'])

    def _network(self, batch):
        """PSMNET.forward without the loss (psmnet/mainmodel.py:67-97)."""
        opt, m = self.option, self.option.model
        a, b = 'left', 'right'
        if 'groupname' in batch and not self.training:
            if batch['groupname'][0] == '2020-2-9_group20':
                a, b = 'right', 'left'
        elif opt.dataset.flip_lr:
            a, b = 'right', 'left'
        ref = self._features(batch[a])
        tar = self._features(batch[b])
        groups = int(m.group_num) if m.cost_volume == 'gwcnet' else 0
        if m.cost_volume not in ('psmnet', 'gwcnet'):
            raise NotImplementedError('cost volume style is not defined : %s' % m.cost_volume)
        vol = ops.psm_volume(ref, tar, [int(d) for d in self.costrange], groups)          # int() truncation, SURVEY Q14
        logits, costs = self._aggregate(vol)
        _, pred_all, prob_all = ops.softargmin_heads(logits, self.disp_values, 4, True)
        return {'pred_depth': pred_all, 'prob_depth': prob_all, 'ref_feature': ops.channel_max(ref),
                '_taps': {'fea_ref': ref, 'fea_tar': tar, 'volume': vol, 'out3': costs[0]}}
