"""StereoDPNet on MI355X: the reference's plugin surface over the libdpf_hip kernels.

Mirrors ``STEREODPNET`` of the reference (src/model/stereodpnet/mainmodel.py:21-177): same constructor
argument (the ``option`` object built from config_/*.json + src/model/stereodpnet/config.json +
dataloader/FaceDP/config.json), same ``forward(batch) -> dict`` keys, same ``state_dict`` key names and
shapes (511 entries + the lazily registered ``normal_estimator.grid``), same loss/metric hooks and
optimiser/scheduler selection.  The implementation is not a module tree of torch layers: parameters live
in ONE flat HBM arena (one fused Adam launch, one RCCL all-reduce over the matching flat gradient arena)
and the forward pass is a straight-line program over the HIP operator layer (ops.py).
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY, ACT_SIGMOID
from .sampler_tables import build_phase_tables, build_shift_tables, is_fractional

try:                                    # optional: neither is installed on the MI355X image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                       # pragma: no cover - depends on the environment
    class _Base(nn.Module):
        """nn.Module with the LightningModule methods the reference's class touches."""

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state_dict names."""


def _attach(root, dotted, tensor, is_param, requires_grad=True):
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if is_param:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad))
    else:
        mod.register_buffer(parts[-1], tensor)


class _Spec(object):
    """Ordered list of (name, shape, kind, init) for every parameter / buffer of the model."""

    def __init__(self):
        self.items = []

    def add(self, name, shape, kind, init):
        self.items.append((name, tuple(shape), kind, init))

    # --- layer helpers (names follow the reference's module tree) ---
    def conv(self, p, cout, cin, ks, bias=None, transpose=False):
        taps = 1
        for k in ks:
            taps *= k
        shape = (cin, cout) + tuple(ks) if transpose else (cout, cin) + tuple(ks)
        self.add(p + '.weight', shape, 'param', ('normal', math.sqrt(2.0 / (taps * cout))))
        if bias is not None:
            self.add(p + '.bias', (cout,), 'param', bias)

    def bn(self, p, c):
        self.add(p + '.weight', (c,), 'param', ('const', 1.0))
        self.add(p + '.bias', (c,), 'param', ('const', 0.0))
        self.add(p + '.running_mean', (c,), 'buffer', ('const', 0.0))
        self.add(p + '.running_var', (c,), 'buffer', ('const', 1.0))
        self.add(p + '.num_batches_tracked', (), 'counter', None)

    def prelu(self, p):
        self.add(p + '.weight', (1,), 'param', ('const', 0.05))

    def convbn2(self, p, cin, cout):
        self.conv(p + '.0', cout, cin, (3, 3))
        self.bn(p + '.1', cout)

    def convbn3(self, p, cin, cout):
        self.conv(p + '.0', cout, cin, (3, 3, 3))
        self.bn(p + '.1', cout)

    def dpblock(self, p, c, t):
        """DPBlock (modules.py:21-35)."""
        for n in ('conv1', 'conv2'):
            self.convbn2(p + '.%s.0' % n, c, c)
            self.prelu(p + '.%s.1' % n)
        for i in range(3):
            self.convbn2(p + '.conv_dilate.%d' % i, c, c)
        self.convbn2(p + '.conv3', 3 * c, c)
        self.convbn2(p + '.conv4.0', c, t * c)
        self.prelu(p + '.conv4.1')
        self.conv(p + '.conv5.depthwise', t * c, 1, (3, 3))
        self.conv(p + '.conv5.pointwise', t * c, t * c, (1, 1))
        self.bn(p + '.conv5.bn', t * c)
        self.prelu(p + '.conv5.prelu')
        self.conv(p + '.conv_skip', t * c, c, (1, 1), bias=('uniform', 1.0 / math.sqrt(c)))
        self.prelu(p + '.prelu')

    def hourglass(self, p, c):
        """PSMNetHourglass (modules.py:204-227)."""
        self.convbn3(p + '.conv1.0', c, 2 * c)
        self.convbn3(p + '.conv2', 2 * c, 2 * c)
        self.convbn3(p + '.conv3.0', 2 * c, 2 * c)
        self.convbn3(p + '.conv4.0', 2 * c, 2 * c)
        self.conv(p + '.conv5.0', 2 * c, 2 * c, (3, 3, 3), transpose=True)
        self.bn(p + '.conv5.1', 2 * c)
        self.conv(p + '.conv6.0', c, 2 * c, (3, 3, 3), transpose=True)
        self.bn(p + '.conv6.1', c)


def build_spec(opt):
    m = opt.model
    c = m.inplanes
    s = _Spec()
    fe = 'feature_extraction'
    # feature_extraction (modules.py:56-91)
    s.convbn2(fe + '.firstconv.0', m.input_channel, c)
    s.convbn2(fe + '.firstconv.2', c, c)
    s.convbn2(fe + '.firstconv.4', c, c)
    s.dpblock(fe + '.block1', c, 1)
    for i in range(m.block_stack):
        s.dpblock(fe + '.interblock1.%d' % i, c, 1)
    s.dpblock(fe + '.block2', c, 2)
    for i in range(m.block_stack):
        s.dpblock(fe + '.interblock2.%d' % i, 2 * c, 1)
    s.dpblock(fe + '.block3', 2 * c, 2)
    for i, cin in enumerate((c, 2 * c, 4 * c)):
        s.conv(fe + '.fpn.inner_blocks.%d' % i, c, cin, (1, 1), bias=('const', 0.0))
    for i in range(3):
        s.conv(fe + '.fpn.layer_blocks.%d' % i, c, c, (3, 3), bias=('const', 0.0))
    # torchvision registers inner/layer blocks interleaved per ModuleList: inner_blocks.* first, then layer_blocks.*
    s.convbn2(fe + '.lastconv.0', 3 * c, 2 * c)
    s.convbn2(fe + '.lastconv.2', 2 * c, c)
    # cost_volume.attention_layer (asm.py:131-146); `normalize` is registered twice (alias keys)
    at = 'cost_volume.attention_layer'
    s.add(at + '.normalize.weight', (c,), 'param', ('const', 1.0))
    s.add(at + '.normalize.bias', (c,), 'param', ('const', 0.0))
    s.conv(at + '.mask_convs.0', c, c, (1, 3, 3))
    s.bn(at + '.mask_convs.1', c)
    s.conv(at + '.mask_convs.3.0', c, c, (1, 1, 1))
    s.add(at + '.mask_convs.3.1.weight', (c,), 'alias', at + '.normalize.weight')
    s.add(at + '.mask_convs.3.1.bias', (c,), 'alias', at + '.normalize.bias')
    # aggregation (modules.py:264-296)
    ag = 'aggregation'
    s.convbn3(ag + '.dres0.0', 2 * c, c)
    s.convbn3(ag + '.dres0.2', c, c)
    s.convbn3(ag + '.dres1.0', c, c)
    s.convbn3(ag + '.dres1.2', c, c)
    for n in ('dres2', 'dres3', 'dres4'):
        s.hourglass(ag + '.' + n, c)
    for n in ('classif1', 'classif2', 'classif3'):
        s.convbn3(ag + '.%s.0' % n, c, c)
        s.conv(ag + '.%s.2' % n, 1, c, (3, 3, 3))
    # normal_estimator (normal_module.py:32-78)
    if m.predict_normal:
        ne = 'normal_estimator'
        s.add(ne + '.costrange', (1, m.level, 1, 1), 'frozen', None)
        if m.use_deform:
            for n, act, cin in (('deform_conv1', 'act1', c + 3), ('deform_conv2', 'act2', 2 * c)):
                fan = cin * 27
                s.add('%s.%s.weight' % (ne, n), (2 * c, cin, 3, 3, 3), 'param', ('uniform', 1.0 / math.sqrt(fan)))
                s.add('%s.%s.bias' % (ne, n), (2 * c,), 'param', ('uniform', 1.0 / math.sqrt(fan)))
                s.conv('%s.%s.conv_offset' % (ne, n), 81, cin, (3, 3, 3), bias=('const', 0.0))   # re-initialised, SURVEY Q4
                s.bn('%s.%s.0' % (ne, act), 2 * c)
        else:
            s.convbn3(ne + '.original_conv.0', c + 3, 2 * c)
            s.convbn3(ne + '.original_conv.2', 2 * c, 2 * c)
        for i, (ci, co) in enumerate(((2 * c, 3 * c), (3 * c, 3 * c), (3 * c, 2 * c), (2 * c, 2 * c), (2 * c, c), (c, 3))):
            s.conv('%s.n_convs.%d.0' % (ne, i), co, ci, (3, 3))
    return s


# left / right feature passes on two HIP streams in training (224.1 -> 221.5 ms per step); DPF_FEATURES_TWO_STREAMS=0: one after the other
FEATURES_TWO_STREAMS = os.environ.get('DPF_FEATURES_TWO_STREAMS', '1') == '1'


class two_stream_grad_warning_off(object):
    """The shared feature extractor's parameters receive gradients from both streams of the two-stream step: intended.  torch's
    accumulate-grad stream-mismatch warning is switched off only around that step (plugin.train_step) and restored afterwards, so a host
    application embedding the plugin keeps the diagnostic for its own models."""

    def __enter__(self):
        self.had = None
        g = torch.autograd.graph
        if FEATURES_TWO_STREAMS and hasattr(g, 'set_warn_on_accumulate_grad_stream_mismatch'):
            probe = getattr(torch._C, '_warn_on_accumulate_grad_stream_mismatch', None)      # the current setting (default: warn)
            self.had = bool(probe()) if callable(probe) else True
            g.set_warn_on_accumulate_grad_stream_mismatch(False)
        return self

    def __exit__(self, *exc):
        if self.had is not None:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(self.had)
        return False


class StereoDPNetCore(_Base):
    """Parameters (flat arena) + the straight-line HIP forward.  ``STEREODPNET`` below adds the plugin hooks."""

    def __init__(self, option):
        super(StereoDPNetCore, self).__init__()
        self.save_hyperparameters()
        self.option = option
        m = option.model
        self.mindisp, self.maxdisp, self.level = m.mindisp, m.maxdisp, m.level
        step = (self.maxdisp / 4.0 - self.mindisp / 4.0) / float(self.level)
        self.costrange = [i * step + self.mindisp / 4.0 for i in range(int(self.level))]          # modules.py:144-145
        n = 4 * int(self.level)
        self.disp_values = [i * ((self.maxdisp - self.mindisp) / float(n)) + self.mindisp for i in range(n)]  # modules.py:345
        self.grid_cache_compat = bool(getattr(m, 'asm_grid_cache_compat', True))
        self._tables = {}
        self._pending_counts = {}
        self.stat_exchange = None          # distributed.StatExchange -> SyncBatchNorm (see enable_sync_batchnorm)
        # mixed precision: the reference's `precision: 16` is PL autocast (every nn.Conv2d / nn.Conv3d in half precision).  Here 16 / 'bf16'
        # make the dense conv kernels round their operands to bf16 (fp32 accumulation, fp32 tensors; ops.conv_operands); 'bf16-2d' is
        # BASELINE configs[4] read literally: only the 2-D convs, on the stand-alone bf16 kernel (conv_bf16.hip).
        prec = str(getattr(option, 'precision', 32))
        self.bf16_all = prec in ('16', 'bf16')
        self.bf16_2d = prec == 'bf16-2d'
        self._build_parameters(self._spec(option))

    @staticmethod
    def _spec(option):
        return build_spec(option)

    def enable_sync_batchnorm(self, group=None):
        """Training BatchNorm statistics over the global batch, like torch.nn.SyncBatchNorm which the reference switches on for
        accelerator == 'ddp' (config_manager.py:57, main.py:55).  ``group=False`` turns it off again."""
        from .distributed import StatExchange
        self.stat_exchange = None if group is False else StatExchange(group)
        return self

    # ------------------------------------------------------------------ parameters
    def _build_parameters(self, spec):
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        total = sum(int(torch.Size(shape).numel()) for _, shape, kind, _ in spec.items if kind == 'param')
        flat = torch.zeros(total, dtype=torch.float32)
        self._layout = []            # (name, offset, numel, shape)
        off = 0
        named = {}
        for name, shape, kind, init in spec.items:
            if kind == 'param':
                numel = int(torch.Size(shape).numel())
                view = flat[off:off + numel].view(shape)
                if init[0] == 'normal':
                    view.normal_(0.0, init[1], generator=g)
                elif init[0] == 'uniform':
                    view.uniform_(-init[1], init[1], generator=g)
                else:
                    view.fill_(init[1])
                self._layout.append((name, off, numel, shape))
                _attach(self, name, view, True)
                named[name] = view
                off += numel
            elif kind == 'buffer':
                _attach(self, name, torch.full(shape, init[1], dtype=torch.float32), False)
            elif kind == 'counter':
                _attach(self, name, torch.zeros((), dtype=torch.long), False)
            elif kind == 'frozen':
                t = torch.tensor(self.costrange, dtype=torch.float32).view(shape)
                _attach(self, name, t, True, requires_grad=False)
        # alias keys: the same Parameter object registered under a second name
        pd = dict(self.named_parameters())
        for name, shape, kind, init in spec.items:
            if kind == 'alias':
                parts = name.split('.')
                mod = self
                for p in parts[:-1]:
                    if p not in mod._modules:
                        mod.add_module(p, _Node())
                    mod = mod._modules[p]
                mod._parameters[parts[-1]] = pd[init]
        self._flat = flat
        self._flat_grad = None
        self._index()

    def _index(self):
        self._P = dict(self.named_parameters(remove_duplicate=False))
        self._B = dict(self.named_buffers())

    def _apply(self, fn, *a, **k):
        super(StereoDPNetCore, self)._apply(fn, *a, **k)
        self._repack()
        return self

    def _repack(self):
        """Re-establish the flat arena after a device / dtype move (Parameter objects are kept)."""
        pd = dict(self.named_parameters())
        dev = pd[self._layout[0][0]].device
        flat = torch.empty(self._flat.numel(), dtype=torch.float32, device=dev)
        for name, off, numel, shape in self._layout:
            p = pd[name]
            flat[off:off + numel].copy_(p.data.reshape(-1))
            p.data = flat[off:off + numel].view(shape)
        self._flat = flat
        self._flat_grad = None
        self._tables = {}
        self._index()

    def flat_parameters(self):
        return self._flat

    def flat_gradients(self, zero=True):
        """Flat gradient arena; every trainable parameter's .grad is a view into it."""
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
            pd = dict(self.named_parameters())
            for name, off, numel, shape in self._layout:
                pd[name].grad = self._flat_grad[off:off + numel].view(shape)
        elif zero:
            self._flat_grad.zero_()
        return self._flat_grad

    def state_dict(self, *a, **k):
        self._flush_counts()
        return super(StereoDPNetCore, self).state_dict(*a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """As nn.Module.load_state_dict; a checkpoint written after the first forward also carries the lazily registered,
        resolution-specific ``normal_estimator.grid`` -- it is materialised here so that resuming into a fresh model
        works under strict=True (the reference's own strict load trips over that key)."""
        key = 'normal_estimator.grid'
        if key in state_dict and 'grid' not in self._modules['normal_estimator']._parameters:
            ref = self._flat
            grid = torch.as_tensor(state_dict[key]).detach().clone().to(device=ref.device, dtype=torch.float32)
            self._modules['normal_estimator'].register_parameter('grid', nn.Parameter(grid, False))
            self._index()
        self._pending_counts = {}
        return super(StereoDPNetCore, self).load_state_dict(state_dict, strict=strict, **kw)

    def _flush_counts(self):
        for name, n in self._pending_counts.items():
            self._B[name] += n
        self._pending_counts = {}

    # ------------------------------------------------------------------ primitives
    def _conv2d(self, *args):
        """nn.Conv2d; with option.precision 'bf16' / 16 on the bf16 MFMA kernel (BASELINE config 5), else exact fp32."""
        return ops.conv2d(*args, bf16=self.bf16_2d)

    def _bn(self, x, p, act=ACT_NONE, slope=None, res=None, res2=None, slope_const=0.0, stats=None):
        P, B = self._P, self._B
        if self.training:
            key = p + '.num_batches_tracked'
            self._pending_counts[key] = self._pending_counts.get(key, 0) + 1
        return ops.norm_act(x, P[p + '.weight'], P[p + '.bias'], slope, res, res2, B[p + '.running_mean'], B[p + '.running_var'],
                            1 if self.training else 2, act, slope_const, self.stat_exchange if self.training else None, stats)

    def _stats_holder(self):
        """conv -> training BatchNorm pairs: the conv's epilogue leaves the channel sums, the BatchNorm skips its statistics pass
        (per-rank statistics only; SyncBatchNorm exchanges {mean, M2} and keeps its own pass)."""
        return {} if (self.training and self.stat_exchange is None) else None

    def _convbn2(self, x, p, stride=1, pad=1, dil=1, act=ACT_NONE, slope=None, res=None):
        st = self._stats_holder()
        y = ops.conv2d(x, self._P[p + '.0.weight'], None, stride, dil if dil > 1 else pad, dil, bf16=self.bf16_2d, stats=st)  # basics.py:17-22
        return self._bn(y, p + '.1', act, slope, res, stats=st)

    def _convbn2_concat(self, x, prefixes, dilations):
        """torch.cat([convbn(x; dilation d) for d], 1) (DPBlock.conv_dilate, modules.py:43-45): every branch's BatchNorm writes its
        channel slice of the concatenated tensor directly."""
        P, B = self._P, self._B
        if self.training and self.stat_exchange is not None:
            # SyncBatchNorm: the three independent branches exchange their statistics in ONE all-gather (and one all-reduce in backward)
            branches = []
            for q, d in zip(prefixes, dilations):
                y = ops.conv2d(x, P[q + '.0.weight'], None, 1, d if d > 1 else 1, d, bf16=self.bf16_2d)
                key = q + '.1.num_batches_tracked'
                self._pending_counts[key] = self._pending_counts.get(key, 0) + 1
                branches.append((y, P[q + '.1.weight'], P[q + '.1.bias'], B[q + '.1.running_mean'], B[q + '.1.running_var'], None))
            return ops.norm_act_concat(branches, 1, ACT_NONE, exchange=self.stat_exchange)
        # DPF_CONV_BN_CAT=1: conv + BatchNorm + cat as ONE autograd node whose backward sums the three data gradients in the transposed-conv
        # epilogue (ops.ConvBnCatFn).  Measured +0.2 % on the step (the epilogue's read-modify-write costs what the two add passes cost), so
        # the default keeps the convolutions as plain launches and only fuses BatchNorm + cat.
        if self.bf16_2d or self.bf16_all or os.environ.get('DPF_CONV_BN_CAT', '0') != '1':
            branches = []
            for q, d in zip(prefixes, dilations):
                y = ops.conv2d(x, P[q + '.0.weight'], None, 1, d if d > 1 else 1, d, bf16=self.bf16_2d)
                if self.training:
                    key = q + '.1.num_batches_tracked'
                    self._pending_counts[key] = self._pending_counts.get(key, 0) + 1
                branches.append((y, P[q + '.1.weight'], P[q + '.1.bias'], B[q + '.1.running_mean'], B[q + '.1.running_var'], None))
            return ops.norm_act_concat(branches, 1 if self.training else 2, ACT_NONE)
        branches = []
        for q in prefixes:
            if self.training:
                key = q + '.1.num_batches_tracked'
                self._pending_counts[key] = self._pending_counts.get(key, 0) + 1
            branches.append((P[q + '.0.weight'], P[q + '.1.weight'], P[q + '.1.bias'], B[q + '.1.running_mean'], B[q + '.1.running_var']))
        # one autograd node: no cat copies, and the three data gradients are summed in the transposed-conv epilogue
        return ops.conv_bn_concat(x, branches, [d if d > 1 else 1 for d in dilations], self.training)

    def _convbn3(self, x, p, stride=1, act=ACT_NONE, res=None):
        st = self._stats_holder()
        y = ops.conv3d(x, self._P[p + '.0.weight'], None, stride, 1, 1, stats=st)                  # basics.py:32-36
        return self._bn(y, p + '.1', act, None, res, stats=st)

    # ------------------------------------------------------------------ feature extractor (modules.py:21-134)
    def _dpblock(self, x, p, s):
        P = self._P
        o1 = self._convbn2(x, p + '.conv1.0', act=ACT_PRELU, slope=P[p + '.conv1.1.weight'])
        o2 = self._convbn2(o1, p + '.conv2.0', act=ACT_PRELU, slope=P[p + '.conv2.1.weight'])
        o2 = self._convbn2_concat(o2, [p + '.conv_dilate.%d' % i for i in range(3)], [2 * i + 1 for i in range(3)])
        o = self._convbn2(o2, p + '.conv3', act=ACT_PRELU, slope=P[p + '.prelu.weight'], res=o1)       # prelu(conv3 + out1)
        o = self._convbn2(o, p + '.conv4.0', s, s, 2, act=ACT_PRELU, slope=P[p + '.conv4.1.weight'])
        d = ops.depthwise_conv3x3(o, P[p + '.conv5.depthwise.weight'])
        d = self._conv2d(d, P[p + '.conv5.pointwise.weight'])
        skip = self._conv2d(x, P[p + '.conv_skip.weight'], P[p + '.conv_skip.bias'], s)
        return self._bn(d, p + '.conv5.bn', ACT_PRELU, P[p + '.conv5.prelu.weight'], None, skip)       # prelu(bn) + skip

    def _features(self, img):
        P, p = self._P, 'feature_extraction'
        x = self._convbn2(img, p + '.firstconv.0', 2, 1, 1, ACT_RELU)
        x = self._convbn2(x, p + '.firstconv.2', act=ACT_RELU)
        x = self._convbn2(x, p + '.firstconv.4', act=ACT_RELU)
        o1 = self._dpblock(x, p + '.block1', 2)
        o2 = o1
        for i in range(self.option.model.block_stack):
            o2 = self._dpblock(o2, p + '.interblock1.%d' % i, 1)
        o2 = self._dpblock(o2, p + '.block2', 2)
        o3 = o2
        for i in range(self.option.model.block_stack):
            o3 = self._dpblock(o3, p + '.interblock2.%d' % i, 1)
        o3 = self._dpblock(o3, p + '.block3', 2)
        # feature pyramid (torchvision FeaturePyramidNetwork semantics; call site modules.py:83-85,119)
        lat = lambda i, t: self._conv2d(t, P['%s.fpn.inner_blocks.%d.weight' % (p, i)], P['%s.fpn.inner_blocks.%d.bias' % (p, i)])
        out = lambda i, t: self._conv2d(t, P['%s.fpn.layer_blocks.%d.weight' % (p, i)], P['%s.fpn.layer_blocks.%d.bias' % (p, i)], 1, 1)
        last = lat(2, o3)
        lo = out(2, last)
        last = ops.nearest_up_add(lat(1, o2), last)
        mid = out(1, last)
        last = ops.nearest_up_add(lat(0, o1), last)
        hi = out(0, last)
        x = ops.concat_channels([hi, ops.upsample_bilinear(mid, 2), ops.upsample_bilinear(lo, 4)])
        x = self._convbn2(x, p + '.lastconv.0', act=ACT_RELU)
        return self._convbn2(x, p + '.lastconv.2', act=ACT_RELU)

    # ------------------------------------------------------------------ cost volume (modules.py:137-200, asm.py)
    def _shift_tables(self, h, w, delta, device):
        key = (h, w, float(delta), str(device))
        if key not in self._tables:
            m = self.option.model
            tables = tuple(t.to(device) for t in build_shift_tables(h, w, delta, m.nearest, m.bilinear, m.phase))
            phase = None
            if is_fractional(delta):                              # per-level shifts only (asm_grid_cache_compat = false)
                phase = tuple(t.to(device) if torch.is_tensor(t) else t for t in build_phase_tables(h, w, delta))
            self._tables[key] = (tables, phase)
        return self._tables[key]

    def _attention_parts(self, fea, delta, stat_sink=None):
        """shifted triple + MaskingAttention up to the sigmoid (asm.py:87-127,141-162).

        ``stat_sink`` = (zeroed mean buffer, zeroed var buffer): the BatchNorm EMA of this call is redirected there
        (it then holds momentum * batch statistic) so the caller can replay the reference's update sequence."""
        P, Bf, p = self._P, self._B, 'cost_volume.attention_layer'
        x3 = ops.shift_triple(fea, *self._shift_tables(fea.shape[2], fea.shape[3], delta, fea.device))  # [B,C,3,h,w]
        st = self._stats_holder()
        mk = ops.conv3d(x3, P[p + '.mask_convs.0.weight'], None, 1, (0, 1, 1), 1, stats=st)
        q = p + '.mask_convs.1'
        if self.training:
            rm, rv = stat_sink if stat_sink is not None else (Bf[q + '.running_mean'], Bf[q + '.running_var'])
            mk = ops.norm_act(mk, P[q + '.weight'], P[q + '.bias'], None, None, None, rm, rv, 1, ACT_RELU, exchange=self.stat_exchange,
                              stats=st)
        else:
            mk = ops.norm_act(mk, P[q + '.weight'], P[q + '.bias'], None, None, None, Bf[q + '.running_mean'], Bf[q + '.running_var'], 2,
                              ACT_RELU)
        mk = ops.conv3d(mk, P[p + '.mask_convs.3.0.weight'])
        s = ops.norm_act(mk, P[p + '.normalize.weight'], P[p + '.normalize.bias'], mode=3, act=ACT_SIGMOID)
        return x3, s

    def _cost_volume(self, ref, tar):
        """CostVolume.build_concat_volume (modules.py:181-197)."""
        L = int(self.level)
        q = 'cost_volume.attention_layer.mask_convs.1'
        if self.grid_cache_compat:
            # The reference's shift-grid cache is keyed on nothing: every level reuses costrange[0], so its
            # 2*L attention calls are L identical (ref, target) pairs.  Compute the pair once, write it to all levels,
            # and replay the 2*L BatchNorm EMA updates in closed form:
            #   r <- 0.9 r + 0.1 s_ref ; r <- 0.9 r + 0.1 s_tar ; ... (L times)
            sink = None
            if self.training:
                C = ref.shape[1]
                z = torch.zeros(4, C, dtype=torch.float32, device=ref.device)
                sink = ((z[0], z[1]), (z[2], z[3]))
            x3f, sf = self._attention_parts(ref, +self.costrange[0], sink[0] if sink else None)
            x3b, sb = self._attention_parts(tar, -self.costrange[0], sink[1] if sink else None)
            if self.training:
                keep = 1.0 - ops.BN_MOMENTUM
                G = sum((keep * keep) ** j for j in range(L))
                for name, a_f, a_b in (('.running_mean', z[0], z[2]), ('.running_var', z[1], z[3])):
                    ops.bn_replay(self._B[q + name], a_f, a_b, keep ** (2 * L), keep * G, G)
                key = q + '.num_batches_tracked'
                self._pending_counts[key] = self._pending_counts.get(key, 0) + 2 * L
            return ops.cv_select(L, [(1 << L) - 1], [x3f, sf, x3b, sb])
        tensors, masks = [], []
        for i, delta in enumerate(self.costrange):
            x3f, sf = self._attention_parts(ref, +delta)
            x3b, sb = self._attention_parts(tar, -delta)
            if self.training:
                key = q + '.num_batches_tracked'
                self._pending_counts[key] = self._pending_counts.get(key, 0) + 2
            tensors += [x3f, sf, x3b, sb]
            masks.append(1 << i)
        return ops.cv_select(L, masks, tensors)

    # ------------------------------------------------------------------ aggregation (modules.py:204-337)
    def _hourglass(self, x, p, presqu, postsqu):
        P = self._P
        out = self._convbn3(x, p + '.conv1.0', 2, ACT_RELU)
        pre = self._convbn3(out, p + '.conv2', 1, ACT_RELU, postsqu)
        out = self._convbn3(pre, p + '.conv3.0', 2, ACT_RELU)
        out = self._convbn3(out, p + '.conv4.0', 1, ACT_RELU)
        up = ops.conv_transpose3d(out, P[p + '.conv5.0.weight'])
        post = self._bn(up, p + '.conv5.1', ACT_RELU, None, presqu if presqu is not None else pre)
        up = ops.conv_transpose3d(post, P[p + '.conv6.0.weight'])
        return up, pre, post

    def _aggregate(self, cost):
        P, p = self._P, 'aggregation'
        c0 = self._convbn3(cost, p + '.dres0.0', 1, ACT_RELU)
        c0 = self._convbn3(c0, p + '.dres0.2', 1, ACT_RELU)
        r = self._convbn3(c0, p + '.dres1.0', 1, ACT_RELU)
        c0 = self._convbn3(r, p + '.dres1.2', 1, ACT_NONE, c0)                        # dres1(cost0) + cost0
        u1, pre1, post1 = self._hourglass(c0, p + '.dres2', None, None)
        o1 = self._bn(u1, p + '.dres2.conv6.1', ACT_NONE, None, c0)                    # conv6 bn + cost0
        u2, _, post2 = self._hourglass(o1, p + '.dres3', pre1, post1)
        o2 = self._bn(u2, p + '.dres3.conv6.1', ACT_NONE, None, c0)
        u3, _, _ = self._hourglass(o2, p + '.dres4', pre1, post2)
        o3 = self._bn(u3, p + '.dres4.conv6.1', ACT_NONE, None, c0)

        def head(x, q):
            y = self._convbn3(x, q + '.0', 1, ACT_RELU)
            return ops.conv3d(y, P[q + '.2.weight'], None, 1, 1, 1)

        k1 = head(o1, p + '.classif1')
        k2 = ops.norm_act(head(o2, p + '.classif2'), res=k1)                           # classif2 + cost1
        k3 = ops.norm_act(head(o3, p + '.classif3'), res=k2)
        if self.training:
            return [k3, k2, k1], [o3, o2, o1]
        return [k3], [o3]

    # ------------------------------------------------------------------ normal module (normal_module.py:140-194)
    def _deform(self, x, p, gi_channels=None):
        P = self._P
        off = ops.conv3d(x, P[p + '.conv_offset.weight'], P[p + '.conv_offset.bias'], 1, 1, 1, gi_channels=gi_channels)
        return ops.deform_conv3d(x, off, P[p + '.weight'], P[p + '.bias'], gi_channels=gi_channels), off

    def _normals(self, cost, disp_full, batch):
        P, p, m = self._P, 'normal_estimator', self.option.model
        B, C, D, h, w = cost.shape
        if 'grid' not in self._modules[p]._parameters:                                  # lazy, resolution specific
            ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
            grid = torch.stack([xs, ys, torch.ones_like(xs)], 0).unsqueeze(0).to(cost.device)
            self._modules[p].register_parameter('grid', nn.Parameter(grid, False))
            self._index()
        if not m.use_sampling:
            raise NotImplementedError('use_sampling=false is not on the StereoDPNet hot path')
        vol, idx = ops.anm_volume(cost, disp_full, batch['K'].float(), batch['abvalue'].float(), self.costrange, int(m.dsample_num),
                                  getattr(self, 'anm_idx_override', None))      # (diagnostic hook of the parity tests, see ops.AnmVolumeFn)
        self.last_anm_idx = idx                  # selected cost levels [B, k, h, w] (diagnostics / tests)
        if m.use_deform:
            # the 3 XYZ channels of `vol` are constants of the batch (no gradient consumer): skip their grad_input
            v1, off1 = self._deform(vol, p + '.deform_conv1', gi_channels=C)
            v1 = self._bn(v1, p + '.act1.0', ACT_RELU)
            v2, off2 = self._deform(v1, p + '.deform_conv2')
            v2 = self._bn(v2, p + '.act2.0', ACT_RELU)
        else:
            v1 = self._convbn3(vol, p + '.original_conv.0', 1, ACT_RELU)
            v2 = self._convbn3(v1, p + '.original_conv.2', 1, ACT_RELU)
            off1 = off2 = None
        Dn = v2.shape[2]
        f = ops.swap_axes12(v2).view(B * Dn, v2.shape[1], h, w)
        for i, dil in enumerate((1, 2, 4, 8, 1, 1)):
            f = self._conv2d(f, P['%s.n_convs.%d.0.weight' % (p, i)], None, 1, dil, dil)
            f = ops.norm_act(f, act=ACT_LEAKY, slope_const=0.1)
        f = ops.upsample_bilinear(f, 4)
        return ops.sigmoid_mean(f, B, Dn), off1, off2

    # ------------------------------------------------------------------ whole network (mainmodel.py:67-104)
    def network(self, batch):
        with ops.conv_operands(self.bf16_all):
            return self._network(batch)

    def _network(self, batch):
        opt = self.option
        a, b = 'left', 'right'
        if 'groupname' in batch and not self.training:
            if batch['groupname'][0] == '2020-2-9_group20':
                a, b = 'right', 'left'
        elif opt.dataset.flip_lr:
            a, b = 'right', 'left'
        if FEATURES_TWO_STREAMS and getattr(self, '_two_streams_ok', False) and self.training and self.stat_exchange is None and batch[a].is_cuda:
            # (enabled by train_step, which sets _two_streams_ok; plain forward() callers stay on one stream)
            # the two feature passes are independent (Q8): the second one runs on its own HIP stream, so that its HBM-bound normalisation
            # kernels overlap the first one's MFMA-bound convolutions (and the other way round); autograd runs each pass's backward on
            # the stream of its forward.  Running statistics stay in the reference's order: every BatchNorm of the second pass waits
            # for the first pass's update of the same layer (ops.BN_ORDER).
            main = torch.cuda.current_stream()
            side = self._feature_stream = getattr(self, '_feature_stream', None) or ops.shared_stream(batch[a].device, 'features')
            events = {}
            side.wait_stream(main)
            ops.BN_ORDER = ('record', events)
            try:
                ref = self._features(batch[a])
                ops.BN_ORDER = ('wait', events)
                # the image was allocated on the caller's stream and is read on `side` -- in forward and again by the first conv's weight
                # gradient in backward: tell the caching allocator, or a caller that drops its batch early gets the block back while that
                # kernel still reads it (seen as a corrupted firstconv.0.0.weight gradient, tools/debug/two_rank_probe.py)
                batch[b].record_stream(side)
                with torch.cuda.stream(side):
                    tar = self._features(batch[b])
            finally:
                ops.BN_ORDER = None
            main.wait_stream(side)
            tar.record_stream(main)
        else:
            ref = self._features(batch[a])
            tar = self._features(batch[b])
        stage = getattr(self, '_grad_stage', None)          # data-parallel step: gradient buckets are exchanged as they complete
        if stage is not None and ref.requires_grad:
            # stage 'aggregation' (cost volume + aggregation stack) is complete once the gradients of BOTH feature maps exist
            left = [2]

            def feat_hook(g):
                left[0] -= 1
                if left[0] == 0:
                    stage('aggregation')
                return g
            ref.register_hook(feat_hook)
            tar.register_hook(feat_hook)
        vol = self._cost_volume(ref, tar)
        logits, costs = self._aggregate(vol)
        preds, pred_all, prob_all = ops.softargmin_heads(logits, self.disp_values, 4, True)
        normal = None
        if opt.model.predict_normal:
            if stage is not None and costs[0].requires_grad:
                # stage 'normal' (normal head): done when the (summed) gradient of its input reaches the aggregation stack
                costs[0].register_hook(lambda g: (stage('normal'), g)[1])
            normal, _, _ = self._normals(costs[0], preds[0], batch)
        return {'pred_depth': pred_all, 'prob_depth': prob_all,
                'pred_normal': normal.unsqueeze(1) if normal is not None else None,
                'ref_feature': ops.channel_max(ref), '_taps': {'fea_ref': ref, 'fea_tar': tar, 'volume': vol, 'out3': costs[0]}}
