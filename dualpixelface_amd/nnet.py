"""NNet ("Normal Assisted Stereo", the reference's src/model/nnet/) behind the same plugin surface:
mainmodel.py:31-177 (NNET), modules.py:45-217 (feature extractor, integer-shift concat volume, regression) and
normal_module_.py:14-117 (the plain, non-deformable normal module) on the HIP operator layer.

Shared with PSMNet: the ResNet/SPP feature extractor (here with half-pixel bilinear resizing of the pyramid branches,
nnet/modules.py:110-120) and the integer-shift cost volume.  Specific: a residual stack of plain 3-D convs instead of hourglasses, a
per-level 2-D refinement of the cost slices guided by the reference features, and the normal module: camera-space coordinate volume
+ both aggregation features -> 3-D convs -> three depth-halving (2,3,3) convs -> dilated 2-D stack -> unit normals.
"""
import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU
from .psmnet import PSMNetCore, build_psmnet_spec
from .stereodpnet import _Spec

REFINE = ((1, 4, 1), (4, 4, 2), (4, 4, 4), (4, 3, 8), (3, 2, 16), (2, 1, 1))      # (in, out) in units of inplanes, dilation; mainmodel.py:50-58
NORMAL = ((1, 3, 1), (3, 3, 2), (3, 3, 4), (3, 2, 8), (2, 2, 16), (2, 1, 1))      # normal_module_.py:36-44


def build_nnet_spec(opt):
    m = opt.model
    c = m.inplanes
    s = _Spec()
    full = build_psmnet_spec(_PsmView(opt))                # the feature extractor's entries, in the reference's registration order
    s.items = [it for it in full.items if it[0].startswith('feature_extraction.')]
    for i, (ci, co, _) in enumerate(REFINE):
        s.conv('convs.%d.0' % i, co * c, ci * c + (1 if i == 0 else 0), (3, 3))
    s.conv('convs.6.0', 1, c, (3, 3))
    s.convbn3('dres0.0', 2 * c, c)
    s.convbn3('dres0.2', c, c)
    for n in ('dres1', 'dres2', 'dres3', 'dres4'):
        s.convbn3(n + '.0', c, c)
        s.convbn3(n + '.2', c, c)
    s.convbn3('classify.0', c, c)
    s.conv('classify.2', 1, c, (3, 3, 3))
    if m.predict_normal:
        nm = 'normal_module'
        s.convbn3(nm + '.wc0.0', 2 * c + 3, c)
        s.convbn3(nm + '.wc0.2', c, c)
        for n in ('pool1', 'pool2', 'pool3'):
            s.conv('%s.%s.0.0' % (nm, n), c, c, (2, 3, 3))
            s.bn('%s.%s.0.1' % (nm, n), c)
        for i, (ci, co, _) in enumerate(NORMAL):
            s.conv('%s.n_convs.%d.0' % (nm, i), co * c, ci * c, (3, 3))
        s.conv(nm + '.n_convs.6.0', 3, c, (3, 3))
        s.add(nm + '.costrange', (1, m.level, 1, 1), 'frozen', None)
    return s


class _PsmView(object):
    """option view for build_psmnet_spec: NNet's config has no cost_volume style key."""

    def __init__(self, opt):
        self.model = _Model(opt.model)


class _Model(object):
    def __init__(self, m):
        self.__dict__.update(m.__dict__)
        self.cost_volume = 'psmnet'
        self.group_num = 0


class NNetCore(PSMNetCore):
    spp_align_corners = False       # nnet/modules.py:110-120

    @staticmethod
    def _spec(option):
        return build_nnet_spec(option)

    def _residual3(self, x, p):
        """dresN(x) + x (mainmodel.py:66-80,136-139): convbn_3d - ReLU - convbn_3d, plus the input."""
        r = self._convbn3(x, p + '.0', 1, ACT_RELU)
        return self._convbn3(r, p + '.2', 1, ACT_NONE, x)

    def _dilated_stack(self, f, p, table, last_out):
        """Sequential of convtext blocks (nnet/modules.py:37-42): 3x3 conv (dilation d, no bias) + LeakyReLU(0.1)."""
        P = self._P
        for i, (_, _, dil) in enumerate(table):
            f = self._conv2d(f, P['%s.%d.0.weight' % (p, i)], None, 1, dil, dil)
            f = ops.norm_act(f, act=ACT_LEAKY, slope_const=0.1)
        return self._conv2d(f, P['%s.6.0.weight' % p], None, 1, 1, 1)            # its LeakyReLU is applied by the caller

    def _refine(self, ref, costs):
        """mainmodel.py:144-147: every cost slice is refined by `convs` on [ref_fea, slice] and added back.  The 8 slices share the
        weights and there is no BatchNorm in `convs`, so they run as one batch of B * level images."""
        B, _, L, h, w = costs.shape
        C = ref.shape[1]
        slices = costs.reshape(B * L, 1, h, w)                                    # [B, 1, L, h, w] -> batch index b * L + level
        guide = ref.unsqueeze(1).expand(B, L, C, h, w).reshape(B * L, C, h, w)
        f = self._dilated_stack(ops.concat_channels([guide, slices]), 'convs', REFINE, 1)
        f = ops.norm_act(f, act=ACT_LEAKY, slope_const=0.1, res2=slices)          # LeakyReLU(conv) + costt
        return f.reshape(B, 1, L, h, w)

    def _normals(self, cost_in0, cost0, batch):
        """NormalModule.forward (normal_module_.py:89-117)."""
        P, p, m = self._P, 'normal_module', self.option.model
        B, C, L, h, w = cost0.shape
        if 'grid' not in self._modules[p]._parameters:                             # lazily registered by the reference (:60-69)
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
            grid = torch.stack([xs, ys, torch.ones_like(xs)], 0).unsqueeze(0).to(cost0.device)
            self._modules[p].register_parameter('grid', nn.Parameter(grid, False))
            self._index()
        # (constant of the model and the batch shape: built once -- a host-to-device copy per step would also keep the step out of a HIP graph)
        lk = (str(cost0.device), B, L, h, w)
        if getattr(self, '_levels_key', None) != lk:
            self._levels = torch.tensor(self.costrange, dtype=torch.float32, device=cost0.device).view(1, L, 1, 1).expand(B, L, h, w).contiguous()
            self._levels_key = lk
        levels = self._levels
        xyz = torch.empty((B, 3, L, h, w), dtype=torch.float32, device=cost0.device)
        ops.xyz_volume_into(xyz, 0, levels, batch['K'].float(), batch['abvalue'].float())
        wc = ops.concat_channels([xyz, cost_in0, cost0])                            # [B, 3 + 2C, L, h, w]
        wc = self._convbn3(wc, p + '.wc0.0', 1, ACT_RELU)
        wc = self._convbn3(wc, p + '.wc0.2', 1, ACT_RELU)
        for n in ('pool1', 'pool2', 'pool3'):                                      # depth 8 -> 4 -> 2 -> 1
            st = self._stats_holder()
            y = ops.conv3d(wc, P['%s.%s.0.0.weight' % (p, n)], None, (2, 1, 1), (0, 1, 1), 1, stats=st)
            wc = self._bn(y, '%s.%s.0.1' % (p, n), ACT_RELU, stats=st)
        D = wc.shape[2]
        f = ops.swap_axes12(wc).reshape(B * D, wc.shape[1], h, w) if D > 1 else wc.reshape(B, wc.shape[1], h, w)
        f = self._dilated_stack(f, p + '.n_convs', NORMAL, 3)
        f = ops.norm_act(f, act=ACT_LEAKY, slope_const=0.1)
        if D > 1:
            f = f.view(B, D, 3, h, w).sum(1)                                        # nmap += slice (:107-110)
        f = ops.upsample_bilinear(f, 4)
        return ops.l2_normalize(f)

    def _network(self, batch):
        """NNET.forward without the loss (mainmodel.py:112-167)."""
        opt, m = self.option, self.option.model
        a, b = 'left', 'right'
        if 'groupname' in batch and not self.training:
            if batch['groupname'][0] == '2020-2-9_group20':
                a, b = 'right', 'left'
        elif opt.dataset.flip_lr:
            a, b = 'right', 'left'
        ref = self._features(batch[a])
        tar = self._features(batch[b])
        vol = ops.psm_volume(ref, tar, [int(d) for d in self.costrange], 0)         # int() truncation (nnet/modules.py:176-178)
        c0 = self._convbn3(vol, 'dres0.0', 1, ACT_RELU)
        cost_in0 = self._convbn3(c0, 'dres0.2', 1, ACT_RELU)
        c = cost_in0
        for n in ('dres1', 'dres2', 'dres3', 'dres4'):
            c = self._residual3(c, n)
        costs = ops.conv3d(self._convbn3(c, 'classify.0', 1, ACT_RELU), self._P['classify.2.weight'], None, 1, 1, 1)
        costss = self._refine(ref, costs)
        _, pred_all, prob_all = ops.softargmin_heads([costs, costss], self.disp_values, 4, align_corners=False)     # mainmodel.py:150-153
        normal = self._normals(cost_in0, c, batch) if m.predict_normal else None
        return {'pred_depth': pred_all, 'prob_depth': prob_all,
                'pred_normal': normal.unsqueeze(1) if normal is not None else None,
                'ref_feature': ops.channel_max(ref),
                '_taps': {'fea_ref': ref, 'fea_tar': tar, 'volume': vol, 'cost0': c, 'costs': costs, 'costss': costss}}
