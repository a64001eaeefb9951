"""StereoNet (the reference's src/model/stereonet/) behind the same plugin surface: mainmodel.py:30-150
(STEREONET) and modules.py:10-120 (FeatureExtraction, EdgeAwareRefinement, disp_regression) on the HIP operator layer.

1/8-resolution features from three 5x5 stride-2 convs and six residual blocks, a *difference* volume over 8 integer shifts,
four 3-D conv-BN-LeakyReLU filters, soft-argmin over the 8 levels, and one edge-aware refinement at full resolution guided by
the right image.  The reference's BasicBlock never calls its `conv2` (modules.py:19-27): those parameters exist (state_dict
contract) and stay untouched.
"""
import math

from . import ops
from .ops import ACT_LEAKY, ACT_RELU
from .stereodpnet import StereoDPNetCore, _Spec

ASTROUS = (1, 2, 4, 8, 1, 1)          # modules.py:66


def build_stereonet_spec(opt):
    m = opt.model
    s = _Spec()
    fe = 'feature_extraction'
    cin = m.input_channel
    for i in range(int(m.k)):
        s.conv('%s.downsample.%d' % (fe, i), 32, cin, (5, 5), bias=('uniform', 1.0 / math.sqrt(cin * 25)))
        cin = 32

    def block(p):
        s.convbn2(p + '.conv1.0', 32, 32)
        s.convbn2(p + '.conv2', 32, 32)

    for i in range(6):
        block('%s.residual_blocks.%d' % (fe, i))
    s.conv(fe + '.conv_alone', 32, 32, (3, 3), bias=('uniform', 1.0 / math.sqrt(32 * 9)))
    for i in range(4):
        s.convbn3('filter.%d.0' % i, 32, 32)
    s.conv('conv3d_alone', 1, 32, (3, 3, 3), bias=('uniform', 1.0 / math.sqrt(32 * 27)))
    er = 'edge_aware_refinements.0'
    s.convbn2(er + '.conv2d_feature.0', 4, 32)
    for i in range(6):
        block('%s.residual_astrous_blocks.%d' % (er, i))
    s.conv(er + '.conv2d_out', 1, 32, (3, 3), bias=('uniform', 1.0 / math.sqrt(32 * 9)))
    return s


class StereoNetCore(StereoDPNetCore):
    def __init__(self, option):
        m = option.model
        # the shared constructor derives costrange / hypothesis values from (mindisp, maxdisp, level): StereoNet's level is 2^k and
        # its regression has one hypothesis per level (mainmodel.py:38-40, modules.py:100-104)
        if not hasattr(m, 'level'):
            m.level = int(math.pow(2, m.k))
        super(StereoNetCore, self).__init__(option)
        L = int(m.level)
        self.disp_values = [i * ((self.maxdisp - self.mindisp) / float(L)) + self.mindisp for i in range(L)]

    @staticmethod
    def _spec(option):
        return build_stereonet_spec(option)

    def _block(self, x, p, dil):
        """BasicBlock.forward (modules.py:19-27): LeakyReLU_0.2(bn(conv(x))) + x -- conv2 is never applied."""
        return self._convbn2(x, p + '.conv1.0', 1, 1, dil, ACT_LEAKY, None, None, slope_const=0.2, res2=x)

    def _convbn2(self, x, p, stride=1, pad=1, dil=1, act=0, slope=None, res=None, slope_const=0.0, res2=None):
        st = self._stats_holder()
        y = ops.conv2d(x, self._P[p + '.0.weight'], None, stride, dil if dil > 1 else pad, dil, bf16=self.bf16_2d, stats=st)
        return self._bn(y, p + '.1', act, slope, res, res2, slope_const, stats=st)

    def _features(self, img):
        """FeatureExtraction.forward (modules.py:53-59)."""
        P, p = self._P, 'feature_extraction'
        x = img
        for i in range(int(self.option.model.k)):
            x = self._conv2d(x, P['%s.downsample.%d.weight' % (p, i)], P['%s.downsample.%d.bias' % (p, i)], 2, 2, 1)
        for i in range(6):
            x = self._block(x, '%s.residual_blocks.%d' % (p, i), 1)
        return self._conv2d(x, P[p + '.conv_alone.weight'], P[p + '.conv_alone.bias'], 1, 1, 1)

    def _refine(self, low, rgb):
        """EdgeAwareRefinement.forward (modules.py:75-93); low [B, h, w], rgb [B, 3, H, W] -> [B, H, W]."""
        P, p = self._P, 'edge_aware_refinements.0'
        H, W = rgb.shape[2], rgb.shape[3]
        scale = 8.0 if float(W) / float(low.shape[-1]) >= 1.5 else 1.0
        # bilinear resizing is linear: scaling the small map first equals `twice_disparity *= 8` after it
        up = ops.resize_bilinear((low * scale).unsqueeze(1), H, W, align_corners=False)
        x = self._convbn2(ops.concat_channels([up, rgb]), p + '.conv2d_feature.0', 1, 1, 1, ACT_LEAKY, slope_const=0.2)
        for i, dil in enumerate(ASTROUS):
            x = self._block(x, '%s.residual_astrous_blocks.%d' % (p, i), dil)
        out = self._conv2d(x, P[p + '.conv2d_out.weight'], P[p + '.conv2d_out.bias'], 1, 1, 1)
        return ops.norm_act(out, res=up, act=ACT_RELU).squeeze(1)                      # ReLU(twice_disparity + conv2d_out)

    def _network(self, batch):
        """STEREONET.forward without the loss (mainmodel.py:79-141)."""
        opt = self.option
        a, b = 'left', 'right'
        if 'groupname' in batch and not self.training:
            if batch['groupname'][0] == '2020-2-9_group20':
                a, b = 'right', 'left'
        elif opt.dataset.flip_lr:
            a, b = 'right', 'left'
        ref = self._features(batch[a])
        tar = self._features(batch[b])
        vol = ops.diff_volume(ref, tar, [int(d) for d in self.costrange])
        x = vol
        for i in range(4):
            st = self._stats_holder()
            y = ops.conv3d(x, self._P['filter.%d.0.0.weight' % i], None, 1, 1, 1, stats=st)
            x = self._bn(y, 'filter.%d.0.1' % i, ACT_LEAKY, slope_const=0.2, stats=st)
        logits = ops.conv3d(x, self._P['conv3d_alone.weight'], self._P['conv3d_alone.bias'], 1, 1, 1)       # [B, 1, L, h, w]
        low, prob = ops.softargmin(logits, self.disp_values, 1, True)
        right = batch['right']
        H, W = right.shape[2], right.shape[3]
        refined = self._refine(low, right)
        coarse = ops.resize_bilinear((low * (float(W) / float(low.shape[-1]))).unsqueeze(1), H, W, align_corners=False).squeeze(1)
        return {'pred_depth': ops.stack_dim1([coarse, refined]), 'prob_depth': prob.unsqueeze(1), 'ref_feature': ops.channel_max(ref),
                '_taps': {'fea_ref': ref, 'fea_tar': tar, 'volume': vol, 'logits': logits, 'low': low}}
