"""``STEREODPNET``: the class the reference's ``model_selector`` instantiates
(src/model/model_selector.py:11-15 -> src/model/stereodpnet/mainmodel.py:21), with the same methods PL / main.py call
(mainmodel.py:67-177) plus an MI355X-native ``train_step`` (flat-arena gradients, fused Adam, optional RCCL all-reduce).
"""
import contextlib
import os

import torch

from . import ops
from .losses import loss_selector
from .selectors import metric_selector, optimizer_selector, scheduler_selector
from .nnet import NNetCore
from .psmnet import PSMNetCore
from .stereonet import StereoNetCore
from .stereodpnet import StereoDPNetCore, two_stream_grad_warning_off


class _PluginHooks(object):
    """The methods PL / main.py call on a model plugin (mainmodel.py:67-177), shared by every model family of the build."""

    def _init_hooks(self, option):
        self.loss_model = loss_selector(option)
        self.metric_model = metric_selector(option)
        self._adam = None

    # ---- reference surface -------------------------------------------------------------------------------
    def forward(self, batch):
        return self._behind_replays(lambda: self._forward(batch))

    def _forward(self, batch):
        results = self.network(batch)
        self.last_taps = results.pop('_taps')
        if self.training and 'disp' in batch:
            results.update(self.loss_model.forward(results, batch))
        return results

    def _loaders(self, train, batch_size, shuffle):
        import torch.utils.data as torch_data
        from dataloader.loader_selector import loader_selector     # this repo's (or a user's) data package, CWD-relative
        dataset = loader_selector(self.option, train)
        if hasattr(dataset, 'produce'):                              # device-side FaceDP path: decode threads + preprocessing kernels
            from .facedp import FaceDPBatcher
            return FaceDPBatcher(dataset, batch_size, shuffle=shuffle, workers=self.option.workers, drop_last=False)
        return torch_data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle,
                                     num_workers=self.option.workers, drop_last=False, pin_memory=self.option.pin_memory)

    def train_dataloader(self):
        return self._loaders(True, self.option.batch_size, True)

    def val_dataloader(self):
        return self._loaders(False, 1, False)

    def test_dataloader(self):
        return self._loaders(False, self.option.batch_size, False)

    def training_step(self, batch, batch_idx):
        results = self.forward(batch)
        losses = {}
        for key, val in results.items():
            if key != 'final_loss' and 'loss' in key:
                losses[key] = val
                self.log(key, val, prog_bar=True)
        return {'loss': results['final_loss'], 'log': losses}

    def validation_step(self, batch, batch_idx):
        results = self.forward(batch)
        if 'depth' in batch:
            self.metric_model.forward(results, batch)
        return results

    def validation_epoch_end(self, outputs):
        self.metric_model.viewer()

    def test_step(self, batch, batch_idx):
        return self.validation_step(batch, batch_idx)

    def test_epoch_end(self, outputs):
        if self.option.mode == 'test':
            self.metric_model.viewer()

    def configure_optimizers(self):
        optimizer = optimizer_selector(self.parameters(), self.option)
        scheduler = scheduler_selector(optimizer, self.option)
        return [optimizer], ([scheduler] if scheduler is not None else [])

    # ---- MI355X-native step ------------------------------------------------------------------------------
    def _grad_pairs(self):
        """[(parameter, its view in the flat gradient arena)] in layout order (cached per arena)."""
        fg = self.flat_gradients(zero=False)
        cache = getattr(self, '_pairs_cache', None)
        if cache is None or cache[0] is not fg:
            pd = dict(self.named_parameters())
            cache = (fg, [(pd[name], fg[off:off + numel].view(shape)) for name, off, numel, shape in self._layout])
            self._pairs_cache = cache
        return cache[1]

    def train_step(self, batch, reducer=None, lr=None):
        """forward + loss + backward + (gradient all-reduce) + fused Adam; returns the results dict.

        Single-process steps on a fixed batch shape are captured into ONE HIP graph after two eager warm-up steps and replayed from then on
        (option.step_graph / DPF_STEP_GRAPH, default on): the ~2 400 kernel launches of a step leave the host once, so the launch gaps between
        the many short kernels disappear.  The C ABI never allocates or synchronises, which is what makes the step capturable."""
        if self.option.optim != 'adam':
            raise NotImplementedError('the fused step implements the shipped Adam configuration')
        # (eager when a per-launch profile is being recorded -- ops.PROFILE -- or gradients are exchanged between ranks)
        if reducer is None and ops.PROFILE is None and _graph_enabled(self) and all(v.is_cuda for v in batch.values() if torch.is_tensor(v)):
            return self._graph_step(batch, lr)
        return self._eager_step(batch, reducer, lr)

    # ---- the step as one HIP graph ---------------------------------------------------------------------
    def _graph_step(self, batch, lr):
        self.train()                                       # (a replay runs no Python: the mode flag must not depend on it)
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        from . import stereodpnet as _sdn
        # everything a captured graph has baked in: shapes, the kernel-path switches, the arenas' addresses (a device move re-creates them)
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items())) + (
            bool(ops.deterministic()), ops.CONV_OPERANDS_BF16, ops.f32_matrix_path(), ops.WGRAD_ASYNC, _sdn.FEATURES_TWO_STREAMS, self.flat_parameters().data_ptr(),
            self.flat_gradients(zero=False).data_ptr(), self._adam['m'].data_ptr() if self._adam else 0, self.stat_exchange is None)
        # a few graph states are kept (most recently used last): the last, partial batch of an epoch has its own key and must not throw the
        # main shape's graph away
        states = self.__dict__.setdefault('_graph_states', [])
        st = next((g for g in states if g['key'] == key), None)
        if st is None:
            st = {'key': key, 'calls': 0, 'graph': None}
            states.append(st)
            del states[:-3]
        else:
            states.remove(st)
            states.append(st)
        self._graph_state = st
        if st.get('failed'):
            return self._eager_step(batch, None, lr)
        st['calls'] += 1
        lr = float(lr if lr is not None else self.option.init_lr)
        flat_g = self.flat_gradients(zero=False)
        # The step lives on its OWN stream, warm-up included: autograd's gradient-accumulation nodes remember the stream they were created
        # on, and a capture cannot make the default stream wait for the capturing one.
        ss = getattr(self, '_step_stream', None)
        if ss is None or ss.device != flat_g.device:
            ss = self._step_stream = ops.shared_stream(flat_g.device, 'step')      # one per process, not per model: ops.shared_stream
        cur = torch.cuda.current_stream(flat_g.device)
        if st['calls'] <= 2 or self._adam is None:         # warm-up: lazily created streams, scratch buffers, sampler tables, kernel attributes
            ss.wait_stream(cur)
            with torch.cuda.stream(ss):
                res = self._eager_step(batch, None, lr)
            cur.wait_stream(ss)
            for v in res.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
            return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in res.items()}
        ad = self._adam
        if st['graph'] is None:
            # static inputs (the caller's tensors are copied in before every replay), the hyper-parameter slot, then the capture itself
            st['inputs'] = {k: v.clone() for k, v in tensors.items()}
            st['extra'] = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
            st['hyper'] = torch.zeros(2, dtype=torch.float32, device=flat_g.device)
            counts_before = dict(self._pending_counts)
            step_before = ad['step']
            self._flush_counts()
            graph = torch.cuda.CUDAGraph()
            try:
                torch.cuda.synchronize()
                ops.reset_zero_arenas()
                # (thread-local error mode: the data path's worker threads -- facedp.FaceDPBatcher decodes and preprocesses the next batch on
                # the GPU meanwhile -- make allocator and copy calls of their own, which a global-mode capture takes for violations)
                with torch.cuda.graph(graph, stream=ss, capture_error_mode='thread_local'):
                    try:
                        cap_batch = dict(st['extra'])
                        cap_batch.update(st['inputs'])
                        res = self._eager_step(cap_batch, None, lr, hyper=st['hyper'])
                        st['results'] = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in res.items()}
                        del res
                    except BaseException as e:               # (ending an invalidated capture has crashed the process before the error could surface)
                        import sys
                        import traceback
                        sys.stderr.write('train_step: exception inside the graph capture: %r\n%s\n' % (e, traceback.format_exc()))
                        sys.stderr.flush()
                        raise
                st['counts'] = dict(self._pending_counts)            # BatchNorm call counters one step adds (host-side bookkeeping)
                self._pending_counts = {}
                ad['step'] -= 1                                      # the capture only RECORDED the step: nothing ran
                st['graph'] = graph
            except Exception as e:                                   # capture refused (unsupported call inside): stay eager, say so once
                import warnings
                warnings.warn('train_step: HIP graph capture failed (%s: %s); continuing with eager launches' % (type(e).__name__, e))
                st['failed'] = True
                self._pending_counts = counts_before
                torch.cuda.synchronize()
                # the aborted capture only RECORDED its work: the clearing fill of the zero arenas never ran although their cursors moved
                # on (the warm-up steps' slots are still dirty), and the Adam step counter may have been advanced
                ops.reset_zero_arenas()
                ad['step'] = step_before
                return self._eager_step(batch, None, lr)
        # The replay runs on the dedicated stream, bracketed by explicit event waits in both directions.  Launched into the caller's stream
        # -- the legacy default stream in a plain script -- kernels the caller enqueued right after the replay were observed to start before
        # the graph had finished (bench.py --wgrad-inline: eager steps behind two replays read parameters the replay was still writing: GPU
        # memory faults and hangs in 2 of 5 runs; none in 12 runs with this ordering).
        ss.wait_stream(cur)
        with torch.cuda.stream(ss):
            for k, v in tensors.items():
                if v.data_ptr() != st['inputs'][k].data_ptr():
                    st['inputs'][k].copy_(v, non_blocking=True)
                    v.record_stream(ss)
            ad['step'] += 1
            h0, h1 = ops.adam_hyper(ad['step'], lr)
            st['hyper'][0:1].fill_(float(h0))              # (scalars travel as kernel arguments: no host buffer the next step could overwrite)
            st['hyper'][1:2].fill_(float(h1))
            st['graph'].replay()
        cur.wait_stream(ss)
        for name, n in st['counts'].items():
            self._pending_counts[name] = self._pending_counts.get(name, 0) + n
        return dict(st['results'])                         # (a fresh dict; the tensors are the graph's static buffers, valid until the next step)

    def _behind_replays(self, fn):
        """Once a train step of this model replays as a HIP graph, everything the model launches one by one (an eager step, forward,
        validation) runs on the SAME stream as the replays, bracketed by event waits against the caller's stream.

        Why (tools/debug/graph_eager_alternation.py): with eager launches on another stream, ordered against the replay
        only through `caller.wait_stream(step stream)` -- an event recorded right behind hipGraphLaunch -- a headline-size replay followed by
        this model's eager forward ended in a GPU memory access fault in about one run of three (100 alternations each); with launch
        blocking, with a host wait, or with the eager work on the replay's own stream: none (0 of 80 / 12 / 14 runs).  4 000 trivial
        kernels behind each replay on the other stream never faulted either: what races is the model's own state (parameters, running
        statistics, cached tables) between the graph's tail and the eager kernels, i.e. the cross-stream event does not cover the whole
        graph on this runtime (ROCm 7.0.2 / PyTorch 2.10).  Same-stream order does not depend on it."""
        ss = getattr(self, '_step_stream', None)
        if ss is None or not any(g.get('graph') is not None for g in getattr(self, '_graph_states', ())):
            return fn()
        cur = torch.cuda.current_stream(ss.device)
        if cur == ss:
            return fn()
        ss.wait_stream(cur)
        with torch.cuda.stream(ss):
            out = fn()
        cur.wait_stream(ss)
        for v in (out.values() if isinstance(out, dict) else ()):
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return out

    def _eager_step(self, batch, reducer=None, lr=None, hyper=None):
        if hyper is None and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            return self._behind_replays(lambda: self._eager_step_body(batch, reducer, lr, None))
        return self._eager_step_body(batch, reducer, lr, hyper)

    def _eager_step_body(self, batch, reducer=None, lr=None, hyper=None):
        self.train()
        gscale = 1.0
        if getattr(self, 'gather_grads', True):
            # Gradients are produced as fresh tensors (autograd hands them over without an accumulate kernel when .grad is None) and
            # gathered into the flat arena by one multi-tensor copy; the 14.7 MB all-reduce then runs once over the arena.  The
            # alternative below accumulates into arena views (one add kernel per parameter, ~300 per step) and overlaps bucketed
            # all-reduces with the backward pass through hooks -- an overlap worth < 0.1 % of a 400 ms step.
            flat_g = self.flat_gradients(zero=False)
            pairs = self._grad_pairs()
            for p, _ in pairs:
                p.grad = None
            ops.wgrad_async_begin([p for p, _ in pairs])

            def gather(sel):
                """Copy the gradients of the selected (parameter, arena view) pairs into the arena; afterwards .grad IS the view."""
                views, grads = [], []
                for p, v in sel:
                    if p.grad is v:
                        continue
                    if p.grad is None:
                        v.zero_()
                    else:
                        views.append(v)
                        grads.append(p.grad)
                if views:
                    torch._foreach_copy_(views, grads)
                for p, v in sel:
                    p.grad = v

            staged = reducer is not None and getattr(reducer, 'collectives', reducer.world_size > 1) and getattr(self, 'stage_grads', True)
            if staged:
                # Data-parallel: the network fires self._grad_stage(bucket) from tensor hooks at its bucket boundaries (normal head done;
                # aggregation + cost volume done); that bucket is gathered and its all-reduce enqueued while the backward pass continues.
                reducer.stage_begin()
                main = torch.cuda.current_stream() if flat_g.is_cuda else None

                def on_stage(name):
                    # `name`: 'aggregation' | 'normal' (StereoDPNetCore._network).  A reducer cut differently has no bucket of that name:
                    # nothing is staged then and stage_finish exchanges everything after backward().
                    bi = reducer.stage_of.get(name) if hasattr(reducer, 'stage_of') else None
                    if bi is None:
                        return
                    # The hook may fire on another stream than the step's (the second feature pass has its own): the bucket's gradients
                    # were produced on the main stream (and the weight-gradient side stream), and Adam will read the arena there -- so the
                    # gather and the collective are enqueued on the main stream, after whatever the hook's stream has produced so far.
                    cur = torch.cuda.current_stream() if main is not None else None
                    if main is not None and cur != main:
                        main.wait_stream(cur)
                    with (torch.cuda.stream(main) if main is not None else contextlib.nullcontext()):
                        _attach_deferred(ops.wgrad_async_take(lambda q: reducer.bucket_of(q) == bi))
                        gather([(p, v) for p, v in pairs if reducer.bucket_of(p) == bi])
                        reducer.stage_launch(bi)
                self._grad_stage = on_stage
            self._two_streams_ok = True            # see StereoDPNetCore._network
            try:
                with two_stream_grad_warning_off():
                    results = self.forward(batch)
                    results['final_loss'].backward()
            finally:
                self._grad_stage = None
                self._two_streams_ok = False
            _attach_deferred(ops.wgrad_async_finish())
            gather(pairs)
            if reducer is not None:
                if staged:
                    reducer.stage_finish()
                else:
                    reducer.reduce_all()
                gscale = 1.0 / reducer.world_size
        else:
            flat_g = self.flat_gradients(zero=True)
            if reducer is not None:
                reducer.begin()
            results = self.forward(batch)
            results['final_loss'].backward()
            if reducer is not None:
                reducer.finish()
                gscale = 1.0 / reducer.world_size
        if self._adam is None or self._adam['m'].device != flat_g.device:
            self._adam = {'m': torch.zeros_like(flat_g), 'v': torch.zeros_like(flat_g), 'step': 0}
        st = self._adam
        st['step'] += 1
        if hyper is not None:                              # (graph capture: the step-dependent scalars come from device memory)
            ops.adam_step_hyper(self.flat_parameters(), flat_g, st['m'], st['v'], hyper, 0.9, 0.999, 1e-5, gscale)
        else:
            ops.adam_step(self.flat_parameters(), flat_g, st['m'], st['v'], st['step'], float(lr if lr is not None else self.option.init_lr),
                          0.9, 0.999, 1e-5, gscale)
        return results


def _graph_enabled(model):
    import os
    env = os.environ.get('DPF_STEP_GRAPH')
    if env is not None:
        return env != '0'
    return bool(getattr(model.option, 'step_graph', True))


def _attach_deferred(pairs):
    """Hand the side-stream weight gradients to their parameters.  A weight used twice per step (the shared feature extractor: left and
    right pass) gets two gradients: the second is added in place by ONE multi-tensor launch for all such weights instead of an add kernel
    (and a fresh tensor) per weight."""
    dst, src, busy = [], [], set()
    for p, g in pairs:
        if p.grad is None:
            p.grad = g
            continue
        if id(p) in busy:                                  # a third gradient for the same weight: finish the pending adds first
            torch._foreach_add_(dst, src)
            dst, src, busy = [], [], set()
        dst.append(p.grad)
        src.append(g)
        busy.add(id(p))
    if dst:
        torch._foreach_add_(dst, src)


class STEREODPNET(_PluginHooks, StereoDPNetCore):
    def __init__(self, option):
        StereoDPNetCore.__init__(self, option)
        self._init_hooks(option)


class PSMNET(_PluginHooks, PSMNetCore):
    """src/model/psmnet/mainmodel.py::PSMNET; its validation hooks are no-ops in the reference (mainmodel.py:143-149)."""

    def __init__(self, option):
        PSMNetCore.__init__(self, option)
        self._init_hooks(option)

    def validation_step(self, batch, batch_idx):
        return None

    def validation_epoch_end(self, outputs):
        return None


class NNET(_PluginHooks, NNetCore):
    """src/model/nnet/mainmodel.py::NNET (mainmodel.py:31-240)."""

    def __init__(self, option):
        NNetCore.__init__(self, option)
        self._init_hooks(option)


class STEREONET(_PluginHooks, StereoNetCore):
    """src/model/stereonet/mainmodel.py::STEREONET (mainmodel.py:30-220)."""

    def __init__(self, option):
        StereoNetCore.__init__(self, option)
        self._init_hooks(option)
