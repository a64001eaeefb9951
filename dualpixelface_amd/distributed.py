"""Data-parallel gradient exchange: one process per GPU, RCCL (backend "nccl" on ROCm) over xGMI.

The model keeps all gradients in one flat arena, so the exchange is an all-reduce(SUM) over a few contiguous slices of
that arena, launched from autograd hooks as soon as a slice is complete so it overlaps the rest of the backward pass
(3 670 492 fp32 = 14.7 MB per step; the reference relies on PL's DDP for the same exchange,
main.py:49).  The division by the world size is fused into the Adam kernel (``gscale``).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1, 0
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        # default timeout: a dead rank or a mismatched training collective fails fast (the long wait for rank-0 validation has its own
        # group, see wait_for_rank0)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    make_wait_group()
    return rank, world, local


_wait_group = None


def make_wait_group(hours=4.0):
    """Create the gloo group of wait_for_rank0 NOW -- right after init_process_group, while the ranks are in lock-step.  new_group is a
    collective over the default group: created lazily at the end of the first epoch (rank 0 still validating, possibly for hours) its
    store waits would have had to honour the long timeout, and a rank that died before the first call would have left the others inside
    group construction instead of inside the monitored barrier that is meant to report it."""
    global _wait_group
    if _wait_group is None and dist.is_initialized() and dist.get_world_size() > 1:
        _wait_group = dist.new_group(backend='gloo', timeout=datetime_timeout(hours))
    return _wait_group


def wait_for_rank0(hours=4.0):
    """Host-side rendezvous after rank 0's validation pass: a gloo barrier on a dedicated group with its own long timeout, so the
    training collectives keep the default one and no GPU spins in an RCCL barrier meanwhile.  If rank 0 dies its sockets close and
    the waiting ranks raise (non-zero exit) instead of hanging."""
    global _wait_group
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    make_wait_group(hours)                               # (normally created in init_from_env / Trainer.fit already)
    dist.monitored_barrier(group=_wait_group, timeout=datetime_timeout(hours), wait_all_ranks=False)


def datetime_timeout(hours):
    import datetime
    return datetime.timedelta(hours=hours)


class FlatGradReducer(object):
    """Bucketed all-reduce over a flat gradient arena.

    ``layout``: ordered [(parameter, offset, numel)] covering the arena in FORWARD order; gradients become ready in
    roughly the reverse order, so buckets are cut from the tail.  ``bucket_bounds``: arena offsets that split it.
    """

    def __init__(self, flat_grad, layout, bucket_bounds=None, group=None, force_collectives=None, stage_of=None):
        self.flat = flat_grad
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        # collectives run when there is somebody to talk to -- or when forced (DPF_FORCE_DIST=1 / bench.py --force-dist: a world-size-1
        # process group still sends every bucket through the backend's stream handling; a one-rank all-reduce(SUM) is the identity)
        if force_collectives is None:
            force_collectives = os.environ.get('DPF_FORCE_DIST', '0') == '1'
        self.collectives = self.world_size > 1 or (bool(force_collectives) and dist.is_initialized())
        # named stages of the staged exchange: {'aggregation': bucket, 'normal': bucket} -- only buckets whose slice holds exactly the
        # parameters the network announces under that name (make_reducer checks the layout); anything else waits for stage_finish
        self.stage_of = dict(stage_of or {})
        self.collective_calls = 0
        n = flat_grad.numel()
        bounds = sorted(set([0, n] + list(bucket_bounds or [])))
        self.buckets = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
        self._need = [0] * len(self.buckets)
        self._param_bucket = {}
        for p, off, numel in layout:
            if not p.requires_grad:
                continue
            for bi, (lo, hi) in enumerate(self.buckets):
                if lo <= off < hi:
                    self._param_bucket[id(p)] = bi
                    self._need[bi] += 1
                    break
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p, _, _ in layout if p.requires_grad]
        self._count = None
        self._work = []

    def begin(self):
        self._count = [0] * len(self.buckets)
        self._work = []

    def _launch(self, bi):
        lo, hi = self.buckets[bi]
        if self.collectives:
            self.collective_calls += 1
            self._work.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        if self._count is None:
            return
        bi = self._param_bucket[id(p)]
        self._count[bi] += 1
        if self._count[bi] == self._need[bi]:
            self._launch(bi)

    def finish(self):
        """Launch any bucket whose hooks did not all fire (unused parameters) and wait for the exchange."""
        for bi in range(len(self.buckets)):
            if self._count is not None and self._count[bi] < self._need[bi]:
                self._launch(bi)
        for w in self._work:
            w.wait()
        self._work = []
        self._count = None

    def reduce_all(self):
        """One all-reduce(SUM) per bucket over the (already complete) arena; no hooks involved."""
        if self.collectives:
            self.collective_calls += len(self.buckets)
            work = [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for lo, hi in self.buckets]
            for w in work:
                w.wait()

    # ---- staged exchange for the gather scheme (plugin.train_step): the backward pass tells the reducer when a bucket's gradients
    # all exist (tensor hooks at the bucket boundaries of the network, see StereoDPNetCore._network); the bucket is gathered into the
    # arena and its all-reduce enqueued right away, overlapping the rest of the backward pass.
    def stage_begin(self):
        self._staged = set()
        self._work = []
        self.log = []                                     # ('launch', bucket) / ('backward_done',) in host order (tests)

    def bucket_of(self, p):
        return self._param_bucket.get(id(p))

    def stage_launch(self, bi):
        """Bucket bi sits complete in the arena: enqueue its all-reduce (asynchronous; stage_finish waits).  A bucket number outside the
        reducer's range is refused (ignored: stage_finish exchanges whatever was not staged)."""
        if bi is None or not (0 <= bi < len(self.buckets)) or bi in self._staged:
            return
        self._staged.add(bi)
        self.log.append(('launch', bi))
        self._launch(bi)

    def stage_finish(self):
        self.log.append(('backward_done',))
        for bi in range(len(self.buckets)):
            if bi not in self._staged:
                self.stage_launch(bi)
        for w in self._work:
            w.wait()
        self._work = []

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_flat(flat, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def make_reducer(model, nbuckets=3, force_collectives=None):
    """Reducer for a StereoDPNetCore: buckets cut at the cost-volume and normal-estimator boundaries.  The staged exchange fires NAMED
    stages ('aggregation' = cost volume + aggregation stack, 'normal' = normal head); a name maps to a bucket only if that bucket's
    slice holds exactly the parameters of that part of the network -- with other cuts (nbuckets 1 or 2, custom bounds) the name is
    absent and its gradients travel with stage_finish."""
    flat_g = model.flat_gradients(zero=True)
    pd = dict(model.named_parameters())
    layout = [(pd[name], off, numel) for name, off, numel, _ in model._layout]
    bounds = []
    if nbuckets >= 2:
        for prefix in ('cost_volume', 'normal_estimator'):
            offs = [off for name, off, _, _ in model._layout if name.startswith(prefix)]
            if offs:
                bounds.append(min(offs))
    red = FlatGradReducer(flat_g, layout, bounds[:max(0, nbuckets - 1)], force_collectives=force_collectives)
    red.stage_of = stage_names(model, red)
    return red


def stage_names(model, reducer):
    """{'aggregation': bucket, 'normal': bucket} for the buckets that hold exactly those parts of the network."""
    def part(name):
        if name.startswith('normal_estimator'):
            return 'normal'
        if name.startswith('feature_extraction'):
            return 'features'
        return 'aggregation'                              # cost volume + aggregation stack + heads
    members = {}
    for name, off, _, _ in model._layout:
        for bi, (lo, hi) in enumerate(reducer.buckets):
            if lo <= off < hi:
                members.setdefault(bi, set()).add(part(name))
    out = {}
    for bi, parts in members.items():
        if len(parts) == 1 and next(iter(parts)) in ('aggregation', 'normal'):
            out[next(iter(parts))] = bi
    return out


class StatExchange(object):
    """Cross-rank exchange of BatchNorm statistics (SyncBatchNorm; SURVEY section 8e "second, optional exchange").

    The reference turns torch's SyncBatchNorm on whenever accelerator == 'ddp' (config_manager.py:57, main.py:55).  Here the
    statistics kernels stay local (ops.NormActFn) and only two tiny vectors per BatchNorm call cross the ranks: the packed
    {mean, M2, count} [2C+1] in forward (all-gather) and {sum dz, sum dz*xhat, .} [3C] in backward (all-reduce).
    """

    def __init__(self, group=None):
        self.group = group
        self.world_size = dist.get_world_size(group)

    def all_gather(self, t):
        out = torch.empty((self.world_size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)
        return out

    def all_reduce_sum_(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

