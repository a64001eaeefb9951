"""ORACLE (test infrastructure only) -- CPU restatement of the StereoDPNet train/eval path.

Plain PyTorch-CPU, functional, driven by a flat ``state`` dict that uses the reference's
``state_dict`` key names.  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package never does and fails loudly without its HIP library.

Pinned by tests/test_oracle_golden.py against tests/golden/e2e_*.npz, which were produced by
importing the reference itself (tests/golden/make_golden.py).  Two boundaries are *not* pinned by
the reference: the deformable conv (oracle/dcn3d.py: parity unpinned, known-answer tests only) and
torchvision's FeaturePyramidNetwork (third-party, torchvision 0.6/0.7 per README.md:19-33, not
vendored; restated from its published semantics).

All citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

from .dcn3d import DeformConv3dFn

BN_EPS = 1e-5
BN_MOM = 0.1


class Cfg(object):
    """Model hyper-parameters (src/model/stereodpnet/config.json)."""

    def __init__(self, mindisp=-4, maxdisp=12, level=8, inplanes=32, dsample_num=4,
                 loss_weight=(1.0, 0.7, 0.5), lambdas=(1.0, 1.0), flip_lr=True, use_deform=True, grid_cache_compat=True):
        self.mindisp, self.maxdisp, self.level, self.inplanes = mindisp, maxdisp, level, inplanes
        self.dsample_num, self.loss_weight, self.lambdas = dsample_num, loss_weight, lambdas
        self.flip_lr, self.use_deform = flip_lr, use_deform
        self.grid_cache_compat = grid_cache_compat      # True: the reference as written; False: per-level shifts

    @property
    def costrange(self):
        # modules.py:144-145 / normal_module.py:76-77
        step = (self.maxdisp / 4.0 - self.mindisp / 4.0) / float(self.level)
        return [i * step + self.mindisp / 4.0 for i in range(self.level)]


class StereoDPNetOracle(object):
    def __init__(self, state, cfg=None, training=True):
        self.S = state
        self.cfg = cfg or Cfg()
        self.training = training
        self.taps = {}

    # ------------------------------------------------------------------ primitives
    def bn(self, x, p):
        S = self.S
        y = F.batch_norm(x, S[p + '.running_mean'], S[p + '.running_var'], S[p + '.weight'], S[p + '.bias'],
                         self.training, BN_MOM, BN_EPS)
        if self.training:
            S[p + '.num_batches_tracked'] += 1
        return y

    def convbn2(self, x, p, stride, pad, dil):
        # basics.py:17-22: padding = dilation if dilation > 1 else pad
        y = F.conv2d(x, self.S[p + '.0.weight'], None, stride, dil if dil > 1 else pad, dil)
        return self.bn(y, p + '.1')

    def convbn3(self, x, p, stride):
        y = F.conv3d(x, self.S[p + '.0.weight'], None, stride, 1)   # basics.py:32-36
        return self.bn(y, p + '.1')

    def prelu(self, x, key):
        return F.prelu(x, self.S[key])

    # ------------------------------------------------------------------ feature extractor
    def dpblock(self, x, p, s, t):
        """modules.py:37-52."""
        o1 = self.prelu(self.convbn2(x, p + '.conv1.0', 1, 1, 1), p + '.conv1.1.weight')
        o2 = self.prelu(self.convbn2(o1, p + '.conv2.0', 1, 1, 1), p + '.conv2.1.weight')
        o2 = torch.cat([self.convbn2(o2, p + '.conv_dilate.%d' % i, 1, 2 * i + 1, 2 * i + 1) for i in range(3)], 1)
        o2 = self.convbn2(o2, p + '.conv3', 1, 1, 1)
        o = self.prelu(o2 + o1, p + '.prelu.weight')
        o = self.prelu(self.convbn2(o, p + '.conv4.0', s, s, 2), p + '.conv4.1.weight')
        # depthwise separable (basics.py:39-58)
        S = self.S
        d = F.conv2d(o, S[p + '.conv5.depthwise.weight'], None, 1, 1, 1, o.shape[1])
        d = F.conv2d(d, S[p + '.conv5.pointwise.weight'])
        d = self.prelu(self.bn(d, p + '.conv5.bn'), p + '.conv5.prelu.weight')
        return d + F.conv2d(x, S[p + '.conv_skip.weight'], S[p + '.conv_skip.bias'], s)

    def fpn(self, feats, p):
        """torchvision.ops.FeaturePyramidNetwork (call site modules.py:83-85,119)."""
        S = self.S
        lat = lambda i, x: F.conv2d(x, S['%s.inner_blocks.%d.weight' % (p, i)], S['%s.inner_blocks.%d.bias' % (p, i)])
        out = lambda i, x: F.conv2d(x, S['%s.layer_blocks.%d.weight' % (p, i)], S['%s.layer_blocks.%d.bias' % (p, i)], 1, 1)
        n = len(feats)
        last = lat(n - 1, feats[-1])
        res = [out(n - 1, last)]
        for i in range(n - 2, -1, -1):
            l = lat(i, feats[i])
            last = l + F.interpolate(last, size=l.shape[-2:], mode='nearest')
            res.insert(0, out(i, last))
        return res

    def feature_extraction(self, img):
        """modules.py:93-134."""
        p = 'feature_extraction'
        x = F.relu(self.convbn2(img, p + '.firstconv.0', 2, 1, 1))
        x = F.relu(self.convbn2(x, p + '.firstconv.2', 1, 1, 1))
        x = F.relu(self.convbn2(x, p + '.firstconv.4', 1, 1, 1))
        o1 = self.dpblock(x, p + '.block1', 2, 1)
        o2 = self.dpblock(self.dpblock(o1, p + '.interblock1.0', 1, 1), p + '.block2', 2, 2)
        o3 = self.dpblock(self.dpblock(o2, p + '.interblock2.0', 1, 1), p + '.block3', 2, 2)
        hi, mid, lo = self.fpn([o1, o2, o3], p + '.fpn')
        mid = F.interpolate(mid, scale_factor=2, mode='bilinear', align_corners=True)
        lo = F.interpolate(lo, scale_factor=4, mode='bilinear', align_corners=True)
        x = torch.cat([hi, mid, lo], 1)
        x = F.relu(self.convbn2(x, p + '.lastconv.0', 1, 1, 1))
        return F.relu(self.convbn2(x, p + '.lastconv.2', 1, 1, 1))

    # ------------------------------------------------------------------ cost volume
    @staticmethod
    def shift_triple(src, delta):
        """asm.py:87-127 for a row shift ``delta`` (already signed: forward = +disp, backward = -disp).

        nearest: grid_sample(mode='nearest') with the default align_corners=False on a grid
        normalised with the align_corners=True formula (asm.py:40-41,96);
        bilinear: align_corners=True (asm.py:101-102); phase: 2-D FFT multiplier that depends on
        the row frequency only, inverted with the old ``irfft(onesided=False)`` semantics
        (asm.py:59-75,112-125)."""
        B, C, h, w = src.shape
        dt = torch.float32
        y = torch.arange(0.0, h, dtype=dt) + torch.tensor(float(delta), dtype=dt)
        x = torch.arange(0.0, w, dtype=dt) + torch.tensor(0.0, dtype=dt)
        yv, xv = torch.meshgrid([y, x], indexing='ij')
        xv = xv / (w - 1) * 2.0 - 1.0
        yv = yv / (h - 1) * 2.0 - 1.0
        grid = torch.stack([xv, yv], -1).unsqueeze(0).expand(B, -1, -1, -1).to(src.dtype)
        near = F.grid_sample(src, grid, mode='nearest', align_corners=False)
        bil = F.grid_sample(src, grid, mode='bilinear', align_corners=True)
        dr = torch.tensor(float(delta), dtype=dt) / h
        nr = torch.cat([torch.arange(0.0, math.ceil(h // 2)), torch.arange(-float(h // 2), 0.0)]).to(dt)   # fp32 like asm.py:68-75
        ang = torch.tensor(2.0 * math.pi, dtype=dt) * (dr * nr.view(h, 1).expand(h, w))
        cosv, sinv = torch.cos(ang), torch.sin(ang)
        spec = torch.fft.fft2(src.float())
        fr = spec.real * cosv - spec.imag * sinv
        fi = spec.imag * cosv + spec.real * sinv
        ph = torch.fft.irfft2(torch.complex(fr, fi)[..., : w // 2 + 1], s=(h, w)).to(src.dtype)
        return near, bil, ph

    def attention(self, x):
        """MaskingAttention.forward (asm.py:158-173); x = [B, C, 3, h, w]."""
        S, p = self.S, 'cost_volume.attention_layer'
        m = F.conv3d(x, S[p + '.mask_convs.0.weight'], None, 1, (0, 1, 1))
        m = F.relu(self.bn(m, p + '.mask_convs.1'))
        m = F.conv3d(m, S[p + '.mask_convs.3.0.weight'])
        m = F.instance_norm(m, None, None, S[p + '.normalize.weight'], S[p + '.normalize.bias'], True, 0.1, 1e-5)
        prob = F.softmax(torch.sigmoid(m), dim=2)
        return torch.mean(x * prob, 2)

    def cost_volume(self, ref, tar, grid_cache_compat=None):
        """CostVolume.build_concat_volume (modules.py:181-197).  With ``grid_cache_compat`` the
        shift grid of level 0 is reused for every level, as the reference's un-keyed cache does
        (asm.py:29-30,51-57)."""
        B, C, h, w = ref.shape
        if grid_cache_compat is None:
            grid_cache_compat = self.cfg.grid_cache_compat
        levels = []
        for lvl, disp in enumerate(self.cfg.costrange):
            d = self.cfg.costrange[0] if grid_cache_compat else disp
            fwd = torch.stack(self.shift_triple(ref, +d), 2)
            bwd = torch.stack(self.shift_triple(tar, -d), 2)
            if lvl == 0:
                self.taps['shift_fwd'], self.taps['shift_bwd'] = fwd, bwd
            a_f = self.attention(fwd)
            a_b = self.attention(bwd)
            if lvl == 0:
                self.taps['attn_fwd'], self.taps['attn_bwd'] = a_f, a_b
            levels.append(torch.cat([a_f, a_b], 1))
        return torch.stack(levels, 2).contiguous()

    # ------------------------------------------------------------------ aggregation
    def hourglass(self, x, p, presqu, postsqu):
        """PSMNetHourglass.forward (modules.py:241-260)."""
        S = self.S
        out = F.relu(self.convbn3(x, p + '.conv1.0', 2))
        pre = self.convbn3(out, p + '.conv2', 1)
        pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
        out = F.relu(self.convbn3(pre, p + '.conv3.0', 2))
        out = F.relu(self.convbn3(out, p + '.conv4.0', 1))
        up = self.bn(F.conv_transpose3d(out, S[p + '.conv5.0.weight'], None, 2, 1, 1), p + '.conv5.1')
        post = F.relu(up + (presqu if presqu is not None else pre))
        out = self.bn(F.conv_transpose3d(post, S[p + '.conv6.0.weight'], None, 2, 1, 1), p + '.conv6.1')
        return out, pre, post

    def aggregation(self, cost):
        """PSMNetHGAggregation.forward (modules.py:310-337)."""
        S, p = self.S, 'aggregation'
        c0 = F.relu(self.convbn3(cost, p + '.dres0.0', 1))
        c0 = F.relu(self.convbn3(c0, p + '.dres0.2', 1))
        r = F.relu(self.convbn3(c0, p + '.dres1.0', 1))
        r = self.convbn3(r, p + '.dres1.2', 1)
        self.taps['cost0_pre'] = r
        c0 = r + c0
        o1, pre1, post1 = self.hourglass(c0, p + '.dres2', None, None)
        o1 = o1 + c0
        o2, _, post2 = self.hourglass(o1, p + '.dres3', pre1, post1)
        o2 = o2 + c0
        o3, _, _ = self.hourglass(o2, p + '.dres4', pre1, post2)
        o3 = o3 + c0

        def head(x, q):
            y = F.relu(self.convbn3(x, q + '.0', 1))
            return F.conv3d(y, S[q + '.2.weight'], None, 1, 1)

        k1 = head(o1, p + '.classif1')
        k2 = head(o2, p + '.classif2') + k1
        k3 = head(o3, p + '.classif3') + k2
        up = lambda k: F.interpolate(k, scale_factor=4, mode='trilinear', align_corners=True).squeeze(1)
        if self.training:
            return [up(k3), up(k2), up(k1)], [o3, o2, o1]
        return [up(k3)], [o3]

    def regression(self, logits):
        """disp_regression.forward (modules.py:341-362)."""
        n = 4 * self.cfg.level
        disp = torch.tensor([i * ((self.cfg.maxdisp - self.cfg.mindisp) / float(n)) + self.cfg.mindisp for i in range(n)],
                            dtype=torch.float64).to(logits[0].dtype).view(1, n, 1, 1)
        preds, probs = [], []
        for l in logits:
            pr = F.softmax(l, 1)
            preds.append(torch.sum(pr * disp, 1))
            probs.append(pr)
        return preds, probs

    # ------------------------------------------------------------------ normal head
    def anm_front(self, cost, disp_full, K, abvalue):
        """normal_module.py:154-167 : top-k level sampling + coordinate volume -> [B, C+3, 4, h, w]."""
        cfg = self.cfg
        B, C, D, h, w = cost.shape
        costv = cost.permute(0, 2, 1, 3, 4)                                           # b d c h w
        disp = F.interpolate(disp_full.unsqueeze(1), scale_factor=0.25, mode='nearest') * 0.25
        cr = torch.tensor(cfg.costrange, dtype=torch.float32).view(1, -1, 1, 1).to(cost.dtype)
        diff = torch.abs(cr - disp)
        score = 1.0 / (diff + 1e-6)
        if getattr(self, 'topk_ties', 'torch') == 'cuda':
            # torch.topk leaves the choice among EQUAL scores at the k-th place to its backend.  The reference runs on CUDA, whose top-k
            # (radix select of the k-th value, then everything above it and the elements equal to it in index order) keeps the lowest
            # indices; the CPU backend (std::nth_element) keeps whatever its partition leaves.  'cuda' states the CUDA rule explicitly
            # (stable descending sort); away from exact ties the two agree.
            idx = torch.sort(score, dim=1, descending=True, stable=True)[1][:, :cfg.dsample_num]
        else:
            _, idx = torch.topk(score, k=cfg.dsample_num, dim=1)                      # :130-131
        idx = torch.sort(idx, dim=1)[0]
        sq_cost = torch.gather(costv, 1, idx.unsqueeze(2).expand(-1, -1, C, -1, -1))
        sq_disp = torch.gather(cr.expand(B, D, h, w), 1, idx)
        # grid_maker_3d (:80-118)
        xs = torch.arange(0, w).to(K.dtype)
        ys = torch.arange(0, h).to(K.dtype)
        yg, xg = torch.meshgrid([ys, xs], indexing='ij')
        pix = torch.stack([xg, yg, torch.ones_like(xg)], 0).view(1, 3, h * w).expand(B, -1, -1)
        Kq = K.clone()
        Kq[:, :2, :] = Kq[:, :2, :] / 4.0
        rays = torch.bmm(torch.inverse(Kq), pix).view(B, 3, h, w).to(cost.dtype)
        a = abvalue[:, 1].view(B, 1, 1, 1).to(cost.dtype)                              # geometry.py:35-40 (Q10)
        b = abvalue[:, 0].view(B, 1, 1, 1).to(cost.dtype)
        depth = a / (sq_disp - b)
        depth = torch.where(torch.isnan(depth) | torch.isinf(depth), torch.zeros_like(depth), depth)
        xyz = rays.unsqueeze(2) * depth.unsqueeze(1)                                   # b 3 d h w
        lo = xyz.reshape(B, -1).min(-1)[0].view(B, 1, 1, 1, 1)
        hi = xyz.reshape(B, -1).max(-1)[0].view(B, 1, 1, 1, 1)
        nxyz = (xyz - lo) / (hi - lo + 1e-6)
        vol = torch.cat([sq_cost.permute(0, 2, 1, 3, 4), nxyz], 1).contiguous()        # b (C+3) d h w
        self.taps['anm_idx'] = idx
        return vol

    def deform(self, x, p):
        """DeformConvPack_dv2.forward (deform_conv.py:323-389): offsets from a plain conv3d."""
        S = self.S
        off = F.conv3d(x, S[p + '.conv_offset.weight'], S[p + '.conv_offset.bias'], 1, 1)
        y = DeformConv3dFn.apply(x, off, S[p + '.weight'], S[p + '.bias'], (1, 1, 1), (1, 1, 1), (1, 1, 1))
        return y, off

    def anm(self, cost, disp_full, batch):
        """ANM.forward (normal_module.py:140-194)."""
        S, p = self.S, 'normal_estimator'
        vol = self.anm_front(cost, disp_full, batch['K'], batch['abvalue'])
        self.taps['anm_volume'] = vol
        v1, off1 = self.deform(vol, p + '.deform_conv1')
        self.taps['dcn1_out'], self.taps['dcn1_offset'] = v1, off1
        v1 = F.relu(self.bn(v1, p + '.act1.0'))
        v2, _ = self.deform(v1, p + '.deform_conv2')
        self.taps['dcn2_out'] = v2
        v2 = F.relu(self.bn(v2, p + '.act2.0'))
        B, C, D, h, w = v2.shape
        f = v2.permute(0, 2, 1, 3, 4).reshape(B * D, C, h, w)
        for i, dil in enumerate((1, 2, 4, 8, 1, 1)):                                     # :59-66
            f = F.leaky_relu(F.conv2d(f, S['%s.n_convs.%d.0.weight' % (p, i)], None, 1, dil, dil), 0.1)
        f = torch.sigmoid(F.interpolate(f, scale_factor=4, mode='bilinear', align_corners=True))
        f = f.view(B, D, 3, 4 * h, 4 * w).mean(1)
        return f * 2.0 - 1.0

    # ------------------------------------------------------------------ losses
    def losses(self, pred_depth, pred_normal, batch):
        """loss_selector.py:29-42, smoothL1.py:15-49 ('given' conversion), cosine.py:35-53 (Q11)."""
        cfg = self.cfg
        mask = batch['mask'] > 0
        n = pred_depth.shape[1]
        wts = [1.0] if n == 1 else list(cfg.loss_weight)
        gt = batch['disp']
        sl1 = sum(wts[i] * F.smooth_l1_loss(pred_depth[:, i][mask], gt[mask]) for i in range(n))
        pn = pred_normal.permute(0, 3, 4, 1, 2)[mask]                                    # [M, 1, 3]
        gn = batch['normal'].permute(0, 2, 3, 1)[mask]                                   # [M, 3]
        pn = pn / torch.norm(pn, p=2, dim=-1, keepdim=True).clamp_min(1e-6)
        gn = gn / torch.norm(gn, p=2, dim=-1, keepdim=True).clamp_min(1e-6)
        a = pn[:, 0]
        den = (torch.norm(a, p=2, dim=-1, keepdim=True) * torch.norm(gn, p=2, dim=-1, keepdim=True)).clamp_min(1e-6)
        sim = ((a * gn) / den).clamp(-1.0, 1.0)
        cos = torch.mean(1.0 - sim)
        final = cfg.lambdas[0] * sl1 + cfg.lambdas[1] * cos
        return {'smoothL1_loss': sl1, 'cosine_loss': cos, 'abvalue': batch['abvalue'], 'final_loss': final}

    # ------------------------------------------------------------------ whole model
    def forward(self, batch):
        """STEREODPNET.forward (mainmodel.py:67-111); flip_lr => the right image is the reference view."""
        a, b = ('right', 'left') if self.cfg.flip_lr else ('left', 'right')
        ref = self.feature_extraction(batch[a])
        tar = self.feature_extraction(batch[b])
        self.taps['fea_ref'], self.taps['fea_tar'] = ref, tar
        vol = self.cost_volume(ref, tar)
        self.taps['volume'] = vol
        logits, costs = self.aggregation(vol)
        self.taps['logits'], self.taps['out3'] = logits, costs[0]
        preds, probs = self.regression(logits)
        normal = self.anm(costs[0], preds[0], batch)
        res = {'pred_depth': torch.stack(preds, 1), 'prob_depth': torch.stack(probs, 1),
               'pred_normal': normal.unsqueeze(1), 'ref_feature': ref.max(1)[0]}
        if self.training and 'disp' in batch:
            res.update(self.losses(res['pred_depth'], res['pred_normal'], batch))
        return res


def adam_step(params, grads, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-5):
    """torch.optim.Adam as configured by model_selector.py:33-34 (eps 1e-5, no weight decay)."""
    with torch.no_grad():
        for k in params:
            g = grads[k]
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
            denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
            params[k].addcdiv_(m[k], denom, value=-lr / bc1)
