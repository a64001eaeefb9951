"""Data-parallel path on a GPU (``-m gpu``): two ranks share the one GPU of the test box over the gloo backend (RCCL refuses two
ranks on one device), which still exercises everything that is ours: parameter broadcast, the bucketed flat-gradient reducer
driven by autograd hooks, the fused Adam with the 1/world scale, and the SyncBatchNorm statistic exchange.

With SyncBatchNorm, 2 ranks x 2 samples must reproduce the single-process step on the 4-sample batch (the loss terms are means
over equally sized, fully masked batches, so the rank-mean of the gradients is the global-batch gradient)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 32, 48


def _model():
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import fill_by_recipe
    m = STEREODPNET(load_option())
    fill_by_recipe(m)
    return m.to('cuda').train()


def _full_batch():
    from dualpixelface_amd.recipe import synthetic_batch
    return synthetic_batch(4, H, W, seed=11)


def _worker(rank, world, port, sync_bn, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    from dualpixelface_amd.distributed import init_from_env, make_reducer, broadcast_flat
    torch.cuda.set_device(0)
    init_from_env('gloo')
    model = _model()
    if rank == 1:                                  # broadcast must repair this
        with torch.no_grad():
            model.flat_parameters().mul_(1.5)
    broadcast_flat(model.flat_parameters(), 0)
    reducer = make_reducer(model)
    if sync_bn:
        model.enable_sync_batchnorm()
    full = _full_batch()
    batch = {k: v[2 * rank:2 * rank + 2].cuda() for k, v in full.items()}
    # count the collectives of the step
    counts = {'all_reduce': 0, 'all_gather': 0}
    real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor

    def ar(*a, **k):
        counts['all_reduce'] += 1
        return real_ar(*a, **k)

    def ag(*a, **k):
        counts['all_gather'] += 1
        return real_ag(*a, **k)
    dist.all_reduce, dist.all_gather_into_tensor = ar, ag
    res = model.train_step(batch, reducer, lr=1e-3)
    dist.all_reduce, dist.all_gather_into_tensor = real_ar, real_ag
    torch.cuda.synchronize()
    out[rank] = (model.flat_parameters().detach().cpu(), model.flat_gradients(zero=False).detach().cpu(), float(res['final_loss']),
                 model.state_dict()['feature_extraction.firstconv.0.1.running_mean'].cpu())
    out['log%d' % rank] = (list(reducer.log), dict(counts))
    dist.destroy_process_group()


def _run(sync_bn):
    port = 33500 + (os.getpid() % 2000) + (7 if sync_bn else 0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, sync_bn, out), nprocs=2, join=True)
    _run.logs = (out['log0'], out['log1'])
    return out[0], out[1]


def test_two_ranks_per_rank_batchnorm():
    (p0, g0, l0, _), (p1, g1, l1, _) = _run(False)
    assert torch.equal(g0, g1), 'both ranks must hold the same summed gradient'
    assert torch.equal(p0, p1), 'parameters must stay replicated after the fused Adam step'
    assert l0 != l1                                 # different samples
    # the summed gradient is the sum of the two local gradients (computed here without any exchange)
    full = _full_batch()
    ref = torch.zeros_like(g0)
    for r in range(2):
        m = _model()
        m.flat_gradients(zero=True)
        res = m.forward({k: v[2 * r:2 * r + 2].cuda() for k, v in full.items()})
        res['final_loss'].backward()
        ref += m.flat_gradients(zero=False).detach().cpu()
    scale = ref.abs().max().item()
    assert (g0 - ref).abs().max().item() <= 2e-3 * scale
    # the exchange overlaps the backward pass: the normal head's bucket (2) and the aggregation + cost-volume bucket (1) are
    # all-reduced from tensor hooks BEFORE backward() returns, the feature extractor's (0) after it; 3 collectives per step in all
    for log, counts in _run.logs:
        assert log == [('launch', 2), ('launch', 1), ('backward_done',), ('launch', 0)], log
        assert counts == {'all_reduce': 3, 'all_gather': 0}, counts


def test_two_ranks_sync_batchnorm_equals_single_process_full_batch():
    (p0, g0, l0, rm0), (p1, g1, l1, rm1) = _run(True)
    assert torch.equal(p0, p1) and torch.equal(g0, g1)
    assert torch.allclose(rm0, rm1, rtol=0, atol=1e-7), 'running statistics follow the global batch on every rank'
    m = _model()
    res = m.train_step({k: v.cuda() for k, v in _full_batch().items()}, None, lr=1e-3)
    ref_p = m.flat_parameters().detach().cpu()
    ref_g = m.flat_gradients(zero=False).detach().cpu()
    assert abs(0.5 * (l0 + l1) - float(res['final_loss'])) <= 1e-4 * abs(float(res['final_loss']))
    assert torch.allclose(rm0, m.state_dict()['feature_extraction.firstconv.0.1.running_mean'].cpu(), rtol=1e-4, atol=1e-6)
    # summed over 2 ranks = 2 x the full-batch gradient (each rank's loss is a mean over its half); fp32 + BatchNorm conditioning
    # of this tiny fixture: relative L2
    rel = ((0.5 * g0 - ref_g).norm() / ref_g.norm()).item()
    assert rel <= 1e-1, rel            # a broken exchange (per-rank statistics) gives O(1)
    # Adam normalises the step: parameters move by <= lr, identically up to that tolerance
    assert (p0 - ref_p).abs().max().item() <= 2.5e-3
    # collectives of a SyncBatchNorm step: one all-gather per training BatchNorm exchange in forward and one all-reduce each in backward,
    # plus the 3 gradient buckets.  119 BatchNorm calls per step (45 per feature pass x 2, 2 attention calls, 25 in the aggregation
    # stack, 2 in the normal head); the three dilated branches of a DPBlock are independent and travel in ONE packed exchange (10
    # DPBlock calls: -20) -> 99.  A BatchNorm's statistics are needed before the next layer can run and its gradient sums before its dx,
    # so exchanges of consecutive layers cannot be merged.
    for log, counts in _run.logs:
        assert counts == {'all_gather': 99, 'all_reduce': 102}, counts


def test_bench_under_torchrun_two_ranks():
    """The driver's multi-GPU invocation of bench.py (torch.distributed.run, one rank per GPU) with two ranks folded onto the test
    box's single GPU over gloo: barrier / MAX-over-ranks timing, reducer, one JSON line from rank 0 with the whole-job rate."""
    import json
    import subprocess
    env = dict(os.environ, DPF_DIST_BACKEND='gloo', DPF_ONE_DEVICE='1', PYTHONPATH=ROOT)
    port = 35500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '1', '--height', '128',
           '--width', '192', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 2 and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and abs(d['value'] - 2 * 2 / (d['ms_per_step'] * 2 / 1e3)) < 1e-6 * d['value'] + 1e-9


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2 ...` exactly as the driver may call it from a bare shell (no torchrun wrapper, no RANK in the
    environment): bench.py spawns its own ranks as child processes before touching the GPU and relays rank 0's JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(DPF_DIST_BACKEND='gloo', DPF_ONE_DEVICE='1', PYTHONPATH=ROOT)
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '1', '--height', '128', '--width', '192',
           '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 2 and d['value'] > 0


def _nccl_one_rank_worker(port, out):
    """One rank, backend 'nccl' (= RCCL): the staged reducer's collectives really go through ProcessGroupNCCL -- its own stream, the
    event hand-over from the stream the bucket was gathered on, work.wait() before Adam -- next to the weight-gradient side stream and the
    second feature stream.  A one-rank all-reduce(SUM) is the identity, so every bucket must come back bit for bit."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dualpixelface_amd.distributed import make_reducer
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    try:
        from dualpixelface_amd.recipe import synthetic_batch
        batch = {k: v.cuda() for k, v in synthetic_batch(2, 64, 96, seed=11).items()}
        model = _model()
        reducer = make_reducer(model, force_collectives=True)
        assert reducer.collectives and reducer.world_size == 1 and reducer.stage_of == {'aggregation': 1, 'normal': 2}
        sent = {}
        launch = reducer._launch

        def spy(bi):                                   # what the bucket holds when its all-reduce is enqueued (same stream: after the gather)
            lo, hi = reducer.buckets[bi]
            sent[bi] = reducer.flat[lo:hi].clone()
            launch(bi)
        reducer._launch = spy
        res = model.train_step(batch, reducer)
        torch.cuda.synchronize()
        order = list(reducer.log)
        flat = model.flat_gradients(zero=False)
        same = all(torch.equal(flat[lo:hi], sent[bi]) for bi, (lo, hi) in enumerate(reducer.buckets))
        # the twin without a reducer (no collectives at all): same weights, same batch
        twin = _model()
        res2 = twin.train_step(batch)
        torch.cuda.synchronize()
        g1, g2 = flat.double().cpu(), twin.flat_gradients(zero=False).double().cpu()
        p1, p2 = model.flat_parameters().double().cpu(), twin.flat_parameters().double().cpu()
        ones = torch.ones(1, device='cuda')
        dist.all_reduce(ones)
        out.update(order=order, same=bool(same), nsent=len(sent), calls=reducer.collective_calls, backend=dist.get_backend(),
                   ranks_seen=float(ones.item()), loss=(float(res['final_loss']), float(res2['final_loss'])),
                   grad_rel=float((g1 - g2).norm() / g2.norm()), grad_finite=bool(torch.isfinite(g1).all()),
                   param_maxdiff=float((p1 - p2).abs().max()))
    finally:
        dist.destroy_process_group()


def test_one_rank_nccl_staged_exchange_is_the_identity():
    """no multi-GPU box exists in the build pool, but a world-size-1 'nccl' group on the one GPU runs exactly the path
    that had never run -- dist.all_reduce(async_op=True) issued from autograd-thread tensor hooks next to the side streams."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        out = mgr.dict()
        p = ctx.Process(target=_nccl_one_rank_worker, args=(port, out))
        p.start()
        p.join(600)
        assert p.exitcode == 0, 'nccl worker failed (exit code %r)' % (p.exitcode,)
        out = dict(out)
    assert out['backend'] == 'nccl' and out['ranks_seen'] == 1.0
    assert out['order'] == [('launch', 2), ('launch', 1), ('backward_done',), ('launch', 0)], out['order']
    assert out['calls'] == 3 and out['nsent'] == 3
    assert out['same'], 'a bucket changed on its way through the one-rank all-reduce (stream ordering?)'
    assert out['grad_finite']
    # against the step without any reducer: equal up to the run-to-run noise of the float atomics
    assert abs(out['loss'][0] - out['loss'][1]) <= 1e-5 * abs(out['loss'][1]), out['loss']
    assert out['grad_rel'] <= 2e-3, out['grad_rel']
    assert out['param_maxdiff'] <= 2.1e-4, out['param_maxdiff']       # one Adam step moves a parameter by at most lr = 1e-4 either way


def test_bench_force_dist_one_rank_nccl():
    """bench.py --gpus 1 --force-dist: the JSON line reports the backend, the ranks RCCL saw and 3 gradient collectives per step."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '2', '--warmup', '1', '--batch', '1',
                        '--height', '64', '--width', '96', '--no-cpu-baseline', '--no-detail'], env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['collective_backend'] == 'nccl' and line['rccl_ranks_seen'] == 1 and line['n_gpus'] == 1
    assert abs(line['gradient_collectives_per_step'] - 3.0) < 1e-9, line['gradient_collectives_per_step']
