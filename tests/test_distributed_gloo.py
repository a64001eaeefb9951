"""CPU, world_size 2, gloo: the bucketed flat-arena gradient all-reduce used for data-parallel training."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dualpixelface_amd.distributed import init_from_env, FlatGradReducer, broadcast_flat
    r, w, _ = init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)
    sizes = [(4, 3), (7,), (2, 5), (6,)]
    n = sum(int(torch.Size(s).numel()) for s in sizes)
    flat = torch.randn(n)
    broadcast_flat(flat, 0)
    flat_g = torch.zeros(n)
    params, layout, off = [], [], 0
    for s in sizes:
        k = int(torch.Size(s).numel())
        p = torch.nn.Parameter(flat[off:off + k].view(s))
        p.grad = flat_g[off:off + k].view(s)
        params.append(p)
        layout.append((p, off, k))
        off += k
    red = FlatGradReducer(flat_g, layout, bucket_bounds=[layout[2][1]])
    assert len(red.buckets) == 2
    x = torch.full((3,), float(rank + 1))
    for step in range(2):
        flat_g.zero_()
        red.begin()
        # parameter 3 is unused on purpose: finish() must still reduce its bucket
        loss = (params[0] @ x).sum() * (rank + 1) + params[1].sum() * 2 + (params[2] ** 2).sum()
        loss.backward()
        red.finish()
    out[rank] = (flat.clone(), flat_g.clone())
    dist.destroy_process_group()


def test_flat_grad_reducer_world2():
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    p0, g0 = out[0]
    p1, g1 = out[1]
    assert torch.equal(p0, p1), 'parameters must be broadcast from rank 0'
    assert torch.equal(g0, g1), 'both ranks must hold the same summed gradient'
    # expected SUM over ranks: d/dW0 = (rank+1) * x_rank broadcast over rows
    exp0 = sum((r + 1) * torch.full((4, 3), float(r + 1)) for r in range(2))
    assert torch.allclose(g0[:12].view(4, 3), exp0)
    assert torch.allclose(g0[12:19], torch.full((7,), 4.0))
    assert torch.allclose(g0[19:29], 2 * 2 * p0[19:29])
    assert torch.equal(g0[29:], torch.zeros(6))
