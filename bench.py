#!/usr/bin/env python3
"""bench.py -- StereoDPNet train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = forward + loss + backward + gradient all-reduce (RCCL, N > 1) + fused Adam on one synthetic batch of
`--batch` dual-pixel pairs of `--height` x `--width` per GPU (weak scaling), inputs resident in HBM before the timed
region.  Rank 0 prints ONE JSON line (contract in the task description) including
  "roofline":     the dominant kernel (implicit-GEMM convolution on the fp32 matrix cores) -- algorithmic FLOPs per
                  launch / average launch duration, measured live with HIP events on the launch stream.  `value` is timed with the step's
                  stream overlap on (weight gradients on a side stream, the two feature passes on two streams); kernels of different
                  streams then share the chip, so this block is measured on 3 further steps of the same model run on ONE stream, where an
                  event pair brackets one kernel's undisturbed run (`measured_in`; `--single-stream` times `value` that way too);
                  "families" lists every timed
                  family (dense conv fwd/dgrad, weight gradient, 1x1 convs as their own HBM-bound family, small-K, deformable conv
                  forward / backward) with TFLOP/s or GB/s and, from the committed PMC passes, HBM bytes per step;
  "roofline_hbm": the largest HBM-bound family (normalisation + activation), algorithmic GB/s against 8 TB/s;
  "cpu_baseline": the CPU oracle (oracle/, a PyTorch-CPU restatement of the reference) timed on this host's cores on a
                  bounded sample (one 1x512x768 train step = BASELINE configs[1]'s shape, 16 threads), scaled to 1024x1536
                  samples/s by pixel count (kind "port").
`python bench.py --gpus N` without RANK in the environment starts its own N ranks (child processes) and relays rank 0's line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PIXEL_FWD_BWD = 2.708e6        # SURVEY.md section 8d (fwd 902 764 FLOP/pixel/pair, fwd+bwd = 3x)
PEAK_F32_TFLOPS = 157.3                 # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0               # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def cpu_baseline(threads, H=512, W=768):
    """The CPU oracle (oracle/, a PyTorch-CPU restatement pinned to the reference's golden vectors) on a bounded sample: one
    forward + loss + backward at BASELINE configs[1]'s shape (1 x 512 x 768), scaled to 1024 x 1536 samples/s by pixel count."""
    from dualpixelface_amd.recipe import synthetic_batch
    from oracle import recipe_state
    from oracle.stereodpnet import StereoDPNetOracle
    torch.set_num_threads(threads)
    batch = synthetic_batch(1, H, W, seed=7)
    st = recipe_state()
    orc = StereoDPNetOracle(st, training=True)
    t0 = time.time()
    res = orc.forward(batch)
    res['final_loss'].backward()
    dt = time.time() - t0
    return {'value': (1.0 / dt) * (H * W) / (1024.0 * 1536.0), 'unit': 'samples/s (1024x1536 equivalent)', 'cores': threads,
            'kind': 'port', 'sample': 'one 1x%dx%d forward+loss+backward of the CPU oracle (%.1f s), scaled by pixel count' % (H, W, dt)}


def stage_bench(args):
    """HBM-bound stages in isolation: achieved = algorithmic bytes / HIP-event time, peak = 8 TB/s."""
    from dualpixelface_amd import load_option, ops
    from dualpixelface_amd.plugin import PSMNET, STEREODPNET
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    B, h, w, C, L = args.batch, args.height // 4, args.width // 4, 32, 8
    g = torch.Generator().manual_seed(3)
    ref = torch.randn(B, C, h, w, generator=g).to(dev)
    tar = torch.randn(B, C, h, w, generator=g).to(dev)
    alg_bytes = float(B) * (2 * C + 2 * C * L) * h * w * 4          # read both feature maps once, write the [2C, L] volume once
    if args.workload == 'psm_volume':
        shifts = [int(i * 0.5 - 1.0) for i in range(L)]
        fn = lambda: ops.psm_volume(ref, tar, shifts, 0)
        name = 'PSMNet concat cost volume (BASELINE configs[3])'
    else:
        fix = args.workload == 'cost_volume_fix'
        model = STEREODPNET(load_option(asm_grid_cache_compat=not fix)).to(dev)
        model.train()
        fn = lambda: model._cost_volume(ref, tar)
        name = 'StereoDPNet cost volume: shift triple + masking attention + volume assembly, forward (SURVEY a2-a4)'
        if fix:
            name += ', per-level fractional shifts (asm_grid_cache_compat=false: 16 attention calls instead of 2)'
    with torch.no_grad():
        for _ in range(args.warmup):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    secs = e0.elapsed_time(e1) * 1e-3 / args.steps
    gbs = alg_bytes / secs / 1e9
    print(json.dumps({'metric': 'stage throughput (not the BASELINE metric)', 'workload': name, 'value': B / secs, 'unit': 'samples/s',
                      'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': secs * 1e3, 'dtype': 'f32', 'data': 'synthetic',
                      'config': {'workload': name, 'batch': B, 'height': args.height, 'width': args.width},
                      'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0, 'traffic': None,
                                   'algorithmic_bytes_per_step': alg_bytes}}))


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: start the N ranks (one process per GPU, torch.distributed.run on 127.0.0.1) as
    CHILD processes -- this process has not touched the GPU and never execs -- and relay rank 0's JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in r.stdout.splitlines():
        if line.startswith('{'):
            print(line)
    sys.stdout.flush()
    sys.exit(r.returncode)


def held_clock_mhz(dev):
    """The shader clock the chip holds under the dominant kernel's load: 60 launches of the aggregation stack's 32 -> 32 3x3x3 layer
    (igemm3_x9_kernel, ~1 ms each) on the current stream while ONE lane on a side stream samples the shader cycle counter against the
    100 MHz counter for 30 ms (dpf_debug_clock_probe).  MI355X clocks down under a dense matrix stream (MI355X_MICROARCH.md, DVFS), so an
    issue-bound kernel's ceiling is `FLOP per clock` x THIS clock, not x 2.4 GHz."""
    import ctypes
    from dualpixelface_amd import ops
    from dualpixelface_amd._lib import lib
    x = torch.randn(4, 32, 8, 256, 384, device=dev)
    w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.1
    out4 = torch.zeros(4, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    one = (1, 1, 1)
    for _ in range(5):
        ops.ConvFn.apply(x, w, None, one, one, one)
    torch.cuda.synchronize()
    for i in range(60):
        ops.ConvFn.apply(x, w, None, one, one, one)
        if i == 8:
            lib().call('dpf_debug_clock_probe', ctypes.c_void_p(out4.data_ptr()), 30000, ctypes.c_void_p(side.cuda_stream))
    torch.cuda.synchronize()
    c0, r0, c1, r1 = (int(v) for v in out4.tolist())
    return (c1 - c0) / max(r1 - r0, 1) * 100.0


def dry_run(args, rank, world):
    """Everything `bench.py --gpus N` does around the GPU work, on the host: the ranks exist and see each other, every rank builds the same
    model (parameters broadcast from rank 0), the reducer cuts the flat gradient arena into its 3 named buckets, one staged exchange in the
    order the backward pass fires it ('normal', 'aggregation', then the feature extractor at stage_finish) sums a per-rank arena over the
    ranks, and rank 0 prints the bench line's launch-related fields."""
    from dualpixelface_amd import load_option
    from dualpixelface_amd.distributed import make_reducer, broadcast_flat
    from dualpixelface_amd.plugin import STEREODPNET
    import torch.distributed as dist
    torch.manual_seed(1 + rank)                       # different initial weights per rank: the broadcast must make them equal
    model = STEREODPNET(load_option())
    broadcast_flat(model.flat_parameters(), 0)
    reducer = make_reducer(model) if world > 1 else None
    psum = model.flat_parameters().double().sum().reshape(1)
    ranks_seen, same_weights, summed_ok, log, ncoll = 1, True, True, [], 0
    if world > 1:
        flat_g = model.flat_gradients(zero=True)
        flat_g.fill_(float(rank + 1))
        reducer.stage_begin()
        reducer.stage_launch(reducer.stage_of.get('normal'))
        reducer.stage_launch(reducer.stage_of.get('aggregation'))
        reducer.stage_finish()
        summed_ok = bool((flat_g == float(world * (world + 1) // 2)).all())
        log, ncoll = list(reducer.log), reducer.collective_calls
        ones = torch.ones(1)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        lo, hi = psum.clone(), psum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same_weights = bool(lo.item() == hi.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'train samples/sec, StereoDPNet 1024x1536 DP pair', 'dry_run': True, 'n_gpus': world, 'value': None,
                          'scaling': 'weak', 'ranks_seen': ranks_seen, 'same_weights_on_every_rank': same_weights,
                          'gradient_arena_summed_over_ranks': summed_ok, 'gradient_collectives_per_step': ncoll,
                          'stage_log': [list(e) for e in log], 'buckets': len(reducer.buckets) if reducer else 0,
                          'collective_backend': dist.get_backend() if dist.is_initialized() else None,
                          'config': {'workload': 'StereoDPNet train step, %d x %dx%d synthetic DP pairs per GPU' % (args.batch, args.height, args.width),
                                     'global_batch': args.batch * world, 'parallelism': 'dp%d' % world}}))
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=4, help='samples per GPU (BASELINE configs[2]: 32 over 8 GPUs)')
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=1536)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--wgrad-inline', '--single-stream', dest='wgrad_inline', action='store_true',
                    help='everything on ONE stream in the timed region too (default: weight gradients on a side stream and the two feature '
                         'passes on two streams, where MFMA-bound and HBM-bound kernels overlap; the per-kernel roofline block is always '
                         'measured on one stream)')
    ap.add_argument('--wgrad-async', action='store_true', help='(default behaviour; kept for older command lines)')
    ap.add_argument('--cpu-baseline-only', default=None, metavar='HxW:threads[,threads...]',
                    help='time only the CPU oracle at the given size for each thread count (e.g. 1024x1536:16,128) and exit')
    ap.add_argument('--model', default='stereodpnet', choices=['stereodpnet', 'psmnet', 'nnet', 'stereonet'],
                    help='psmnet = BASELINE configs[3] (cross-model plugin check): the PSMNet plugin on the same kernels; nnet = the NNet plugin')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'bf16-2d'],
                    help="bf16 = BASELINE configs[4]: bf16-operand MFMA for the 2-D convs, fp32 everywhere else (default: exact fp32)")
    ap.add_argument('--sync-bn', action='store_true',
                    help='BatchNorm statistics over the global batch (what the reference does under DDP); default per-rank statistics')
    ap.add_argument('--shapes', default=None, help='write a per-convolution-shape timing table to this file')
    ap.add_argument('--no-detail', action='store_true', help='skip the two extra untimed steps that time the normalisation / activation launches')
    ap.add_argument('--force-dist', action='store_true',
                    help="with --gpus 1: still create a (world-size-1) 'nccl' process group and send every gradient bucket through it -- the "
                         "staged reducer (tensor-hook launches, side-stream joins, stage_finish, fused Adam) runs through RCCL's stream handling")
    ap.add_argument('--no-graph', action='store_true', help='eager kernel launches in the timed region (default: the step replays as one HIP graph)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launch path only: parse, start / join the ranks, build the model and the 3-bucket reducer on the HOST, exchange one '
                         'staged gradient arena over the process group (DPF_DIST_BACKEND=gloo on a box without GPUs), print the line with '
                         '"dry_run": true -- no GPU call is made (tests/test_host_logic.py covers `--gpus 8` this way)')
    ap.add_argument('--workload', default='train', choices=['train', 'psm_volume', 'cost_volume', 'cost_volume_fix'],
                    help="'train' = the BASELINE metric; the other two time one HBM-bound stage in isolation (BASELINE configs[3], SURVEY a2-a4)")
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ and args.workload == 'train':
        return self_launch(args.gpus)
    if args.cpu_baseline_only:
        size, threads = args.cpu_baseline_only.split(':')
        h, w = (int(v) for v in size.split('x'))
        for t in threads.split(','):
            print(json.dumps(cpu_baseline(int(t), h, w)))
        return

    from dualpixelface_amd import load_option, ops
    from dualpixelface_amd.distributed import init_from_env, make_reducer, broadcast_flat
    from dualpixelface_amd.plugin import NNET, PSMNET, STEREODPNET, STEREONET
    from dualpixelface_amd.recipe import synthetic_batch
    import torch.distributed as dist

    if args.workload != 'train':
        return stage_bench(args)
    # test hooks (tests/test_gpu_distributed.py): DPF_DIST_BACKEND=gloo and DPF_ONE_DEVICE=1 let two ranks share the one GPU of a test box
    rank, world, local = init_from_env(os.environ.get('DPF_DIST_BACKEND'))
    if args.force_dist and world == 1 and not dist.is_initialized():
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        torch.cuda.set_device(0)
        dist.init_process_group(os.environ.get('DPF_DIST_BACKEND') or 'nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or without RANK set)' % (args.gpus, world, args.gpus)
    if args.dry_run:
        return dry_run(args, rank, world)
    if os.environ.get('DPF_ONE_DEVICE'):
        local = 0
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    torch.manual_seed(1)
    opt = load_option({'psmnet': 'train_faceDP_psmnet', 'nnet': 'train_faceDP_nnet', 'stereonet': 'train_faceDP_stereonet'}.get(args.model, 'train_faceDP'))
    if args.precision != 'f32':
        opt.precision = args.precision
    model = {'psmnet': PSMNET, 'nnet': NNET, 'stereonet': STEREONET}.get(args.model, STEREODPNET)(opt)     # reference initialisation scheme, random weights
    model.to(dev)
    broadcast_flat(model.flat_parameters(), 0)
    reducer = make_reducer(model, force_collectives=args.force_dist) if (world > 1 or args.force_dist) else None
    if args.sync_bn and world > 1:
        model.enable_sync_batchnorm()
    batch = {k: v.to(dev) for k, v in synthetic_batch(args.batch, args.height, args.width, seed=rank).items()}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import dualpixelface_amd.stereodpnet as sdn

    def set_streams(on):
        ops.WGRAD_ASYNC = bool(on) and os.environ.get('DPF_WGRAD_ASYNC', '1') == '1'
        sdn.FEATURES_TWO_STREAMS = bool(on) and os.environ.get('DPF_FEATURES_TWO_STREAMS', '1') == '1'
    set_streams(not args.wgrad_inline)
    # the step as ONE HIP graph (plugin.train_step: captured on the third call of a shape, replayed afterwards; single-process steps only --
    # with a reducer the collectives stay eager).  The capture must not fall into the timed region: at least four untimed steps.
    use_graph = reducer is None and not args.no_graph and os.environ.get('DPF_STEP_GRAPH', '1') != '0'
    if not use_graph:
        os.environ['DPF_STEP_GRAPH'] = '0'
    for _ in range(max(args.warmup, 4) if use_graph else args.warmup):
        model.train_step(batch, reducer)
    sync()
    graph_live = bool(use_graph and getattr(model, '_graph_state', None) and model._graph_state.get('graph') is not None)
    ops.PROFILE = None if graph_live else []             # per-launch event pairs need eager launches: with the graph live they come from the detail steps only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = model.train_step(batch, reducer)
    sync()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = (ops.PROFILE or []), None
    loss = float(res['final_loss'].detach())
    # The same step with the fp32 products formed by the EXACT construction (six bf16 partial products of exact three-way splits,
    # dpf_set_f32_matrix_path(1)) instead of the default three f16 products of block-scaled splits: a short second measurement (the step
    # graph is re-captured: the matrix path is part of its key), reported beside `value`, never as `value`.
    value_path1 = None
    mpath_default = ops.f32_matrix_path()
    if world == 1 and args.precision == 'f32' and args.model == 'stereodpnet' and mpath_default == 2 and not args.no_detail:
        ops.set_f32_matrix_path(1)
        for _ in range(4):
            model.train_step(batch, reducer)
        sync()
        t1 = time.perf_counter()
        for _ in range(5):
            model.train_step(batch, reducer)
        sync()
        value_path1 = args.batch * 5 / (time.perf_counter() - t1)
        ops.set_f32_matrix_path(mpath_default)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # a few extra, UNTIMED steps with the detail timers on: the 744 normalisation / activation launches of a step are timed here so that
    # their event records cannot perturb the headline number
    # Per-kernel attribution: with the weight gradients on a side stream (the default, and what `value` is measured on) kernels of the
    # two streams overlap and an event pair no longer brackets ONE kernel's undisturbed run.  The roofline block is therefore measured
    # on `detail_steps` further steps of the same model / batch with the weight gradients IN LINE (no overlap: event time == kernel
    # time, equal to the rocprofv3 averages in profiles/), which also carry the detail timers of the 744 normalisation launches.
    prof_timed, timed_async = prof, (ops.WGRAD_ASYNC or sdn.FEATURES_TWO_STREAMS)
    prof_detail, detail_steps = [], 0
    if not args.no_detail:
        set_streams(False)
        for _ in range(2):
            model.train_step(batch, reducer)                      # settling steps in the one-stream mode
        ops.PROFILE, ops.PROFILE_DETAIL, detail_steps = [], True, 3
        for _ in range(detail_steps):
            model.train_step(batch, reducer)
        sync()
        prof_detail, ops.PROFILE, ops.PROFILE_DETAIL = ops.PROFILE, None, False
        # These steps launch kernel by kernel, and an event pair also brackets the time the GPU waits for the host between the launches of
        # one operation.  On a box whose host is slow (observed on fresh boxes: 5 x on the multi-launch normalisation ops) that idle time
        # would be booked as kernel time: every record takes the MINIMUM over the detail steps of the record at the same position of the
        # step (the launch sequence is identical), and the per-step sums go to stderr.
        ms = [r[2].elapsed_time(r[3]) for r in prof_detail]
        per = len(ms) // detail_steps
        if per * detail_steps == len(ms) and all(prof_detail[i][4] == prof_detail[i + per][4] for i in range(0, len(ms) - per, max(per // 50, 1))):
            sys.stderr.write('detail steps, event-time sums in ms: %s\n' % ', '.join('%.1f' % sum(ms[s * per:(s + 1) * per]) for s in range(detail_steps)))
            best = [min(ms[i + s * per] for s in range(detail_steps)) for i in range(per)]
            ms = best * detail_steps

        class _Ms(object):
            def __init__(self, v):
                self.v = v

            def elapsed_time(self, other):
                return self.v
        prof_detail = [(r[0], r[1], _Ms(m), None, r[4], r[5]) for r, m in zip(prof_detail, ms)]
        set_streams(not args.wgrad_inline)
        prof, prof_steps = [r for r in prof_detail if r[0] != 'norm_act'], detail_steps
    else:
        prof_steps = args.steps
    clock_mhz = None
    if not args.no_detail and args.precision == 'f32' and args.model == 'stereodpnet':
        clock_mhz = held_clock_mhz(dev)
    ranks_seen = world
    if world > 1 or (args.force_dist and dist.is_initialized()):
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                  # every rank contributes 1: the sum is the number of ranks RCCL really joined
        ranks_seen = int(round(float(ones.item())))

    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / elapsed

        def families(records, steps):
            fam, shapes = {}, {}
            for family, flops, e0, e1, tag, nbytes in records:
                secs = e0.elapsed_time(e1) * 1e-3
                f = fam.setdefault(family, [0.0, 0.0, 0, 0.0])
                f[0] += flops; f[1] += secs; f[2] += 1; f[3] += nbytes
                g = shapes.setdefault(tag, [0.0, 0.0, 0])
                g[0] += flops; g[1] += secs; g[2] += 1
            return fam, shapes
        fam, shapes = families(prof, prof_steps)
        fam_t, _ = families(prof_timed, args.steps)
        fam_d, _ = families([r for r in prof_detail if r[0] == 'norm_act'], detail_steps)
        if args.shapes:
            with open(args.shapes, 'w') as fh:
                for tag, (fl, se, n) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                    fh.write('%-60s calls %4d  ms/step %8.3f  TFLOP/s %6.1f\n' % (tag, n, se / prof_steps * 1e3, fl / se / 1e12))
        # HBM bytes per step and family from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_profiles.sh).  The file
        # carries a hash of the kernel sources it was collected on; the numbers are only reported for the workload, precision, world size
        # and kernels they were measured with -- otherwise `traffic` is null and `traffic_source` says why.
        tname = 'r06_pmc_traffic_bf16.json' if args.precision == 'bf16' else 'r06_pmc_traffic.json'
        tpath = os.path.join(ROOT, 'profiles', tname)
        pmc = json.load(open(tpath)) if os.path.exists(tpath) else {}
        meta = pmc.pop('_meta', {})
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        try:
            from pmc_traffic import kernel_sources_sha16
            sha_now = kernel_sources_sha16()
        except Exception:
            sha_now = None
        same_workload = args.model == 'stereodpnet' and (args.batch, args.height, args.width) == (4, 1024, 1536) and world == 1 and not args.sync_bn
        same_kernels = bool(meta) and meta.get('kernel_sources_sha16') == sha_now
        if not pmc:
            why = 'no traffic file'
        elif not same_workload:
            why = 'not the workload the counters were collected on (StereoDPNet, 4 x 1024x1536, 1 GPU, per-rank BatchNorm)'
        elif not same_kernels:
            why = 'kernel sources changed since the counters were collected'
        else:
            why = None
        traffic_source = {'file': 'profiles/' + tname, 'kernel_sources_sha16': meta.get('kernel_sources_sha16'), 'collected_utc': meta.get('collected_utc'),
                          'kernel_sources_sha16_now': sha_now, 'used': why is None, 'not_used_because': why}
        if why is not None:
            pmc = {}
            same_workload = False

        def hbm_per_step(*keys):
            recs = [pmc[k] for k in keys if k in pmc]
            return sum(r['hbm_bytes_per_step'] for r in recs) if (recs and same_workload) else None
        PMC_KEYS = {'conv_igemm': ('igemm2',), 'conv_wgrad': ('wgrad2',), 'conv_pointwise': ('pointwise',), 'conv_smallk': ('smallk',),
                    'dcn_fwd': ('dcn_fwd',), 'dcn_bwd': ('dcn_bwd_input', 'dcn_bwd_offset'), 'norm_act': ('bn_',)}
        MFMA_FAMILIES = ('conv_igemm', 'conv_wgrad', 'conv_bf16', 'dcn_fwd', 'dcn_bwd')
        fam_out = {}
        for k, (flops, secs, n, nbytes) in list(fam.items()) + list(fam_d.items()):
            steps = detail_steps if k == 'norm_act' else prof_steps
            rec = {'ms_per_step': secs / steps * 1e3, 'launches_per_step': n / steps}
            if k in MFMA_FAMILIES or k == 'conv_smallk':
                rec['tflops'] = flops / secs / 1e12
            if nbytes > 0:
                rec['algorithmic_gbs'] = nbytes / secs / 1e9
                rec['algorithmic_bytes_per_step'] = nbytes / steps
            hb = hbm_per_step(*PMC_KEYS.get(k, ()))
            if hb is not None:
                rec['hbm_bytes_per_step'] = hb
                if nbytes > 0:
                    rec['traffic_over_algorithmic'] = hb / (nbytes / steps)
            fam_out[k] = rec
        dom = 'conv_igemm' if 'conv_igemm' in fam else (max(fam, key=lambda k: fam[k][1]) if fam else None)
        roof = None
        if dom:
            flops, secs, n, alg_bytes = fam[dom]
            ach = flops / secs / 1e12
            hb = hbm_per_step(*PMC_KEYS.get(dom, ()))
            # the dense conv kernels contract on the bf16 matrix cores in --precision bf16: price them against THAT peak (they are then
            # staging-bound -- LDS-DMA / LDS reads of the fp32 patch -- far below it)
            peak = PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS
            # how precision "f32" multiplies (dpf_get_f32_matrix_path): 0 = fp32 MFMA only, 1 = six bf16 partial products per fp32 product,
            # 2 = three f16 partial products of block-scaled two-way splits -- the ceiling of the split paths is the 16-bit pipe's dense peak
            # divided by the products per fp32 product (x 27/28: 27 taps in 7 groups of 4)
            from dualpixelface_amd._lib import lib as _dpf_lib
            mpath = int(_dpf_lib().cdll.dpf_get_f32_matrix_path())
            nprod = {1: 6.0, 2: 3.0}.get(mpath, 6.0)
            pipe_name = {1: 'six bf16 partial products of exact three-way splits', 2: 'three f16 partial products of block-scaled two-way splits'}.get(mpath, 'fp32 MFMA')
            x6_flop_per_clk = 1024 * (32 * 32 * 16 * 2 / 32.0) / nprod * 27.0 / 28.0     # fp32 FLOP per shader clock of the split construction
            x6_peak = x6_flop_per_clk * 2.4e9 / 1e12
            f32_frac_pipe = None
            if args.precision == 'f32' and dom == 'conv_igemm':
                # The family mixes two pipes: stride-1 launches with >= 8 input channels multiply on the bf16 pipe (six partial products per fp32
                # product: ceiling 2.5 PFLOP/s / 6 x 27/28), the rest (stride 2, the 3-channel first layer) on v_mfma_f32_32x32x2_f32 (157.3).
                # `peak` is the FLOP-weighted (harmonic) blend of the two -- the rate at which the family would run with every launch at ITS
                # pipe's dense peak -- so that `frac` cannot exceed 1 (157.3 is not the peak of a bf16-pipe kernel).
                on_bf16 = lambda tag: mpath > 0 and (' s1 ' in tag) and not any((' C%d ' % c) in tag for c in range(1, 8)) and ' k111 ' not in tag
                fl_b = sum(r[1] for r in prof if r[0] == dom and on_bf16(r[4]))
                fl_f = sum(r[1] for r in prof if r[0] == dom and not on_bf16(r[4]))
                if fl_b + fl_f > 0:
                    peak = (fl_b + fl_f) / (fl_b / x6_peak + fl_f / PEAK_F32_TFLOPS)
                    f32_frac_pipe = fl_b / (fl_b + fl_f)
            inline_ms = sum(v[1] for v in fam.values()) / prof_steps * 1e3
            roof = {'bound': 'mfma', 'kernel': dom, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                    'traffic': (hb / (n / prof_steps)) if hb is not None else None, 'traffic_source': traffic_source, 'algorithmic_bytes_per_launch': alg_bytes / n, 'launches': n,
                    'avg_launch_ms': secs / n * 1e3, 'ms_per_step': secs / prof_steps * 1e3,
                    'measured_in': ('%d steps after the timed region on ONE stream (weight gradients in line, feature passes one after the other: kernels do not overlap, event time = kernel time)' % prof_steps)
                                   if not args.no_detail else 'the timed region', 'families': fam_out}
            if args.precision != 'bf16':
                # (see the blend above: `peak` prices every launch at the dense peak of the pipe it runs on)
                roof['peak_note'] = ('FLOP-weighted blend of the pipes the launches run on: %.0f %% of the family\'s fp32 FLOPs multiply on the 16-bit matrix pipe '
                                     '(%s: 2.5 PFLOP/s / %d x 27/28 = %.1f TFLOP/s of fp32 FLOPs at 2.4 GHz), the rest on '
                                     'v_mfma_f32_32x32x2_f32 (157.3)' % (100.0 * (f32_frac_pipe or 0.0), pipe_name, int(nprod), x6_peak))
                roof['f32_matrix_path'] = mpath
                roof['frac_of_f32_mfma_peak'] = ach / PEAK_F32_TFLOPS        # the dtype's own dense peak (individual bf16-pipe launches exceed it)
                if clock_mhz:
                    # the bf16-pipe share priced at the clock the chip HOLDS under that stream (power-bound: MI355X_MICROARCH.md, DVFS)
                    held_b = x6_flop_per_clk * clock_mhz * 1e6 / 1e12
                    fb = f32_frac_pipe if f32_frac_pipe is not None else 1.0
                    held = 1.0 / (fb / held_b + (1.0 - fb) / PEAK_F32_TFLOPS)
                    roof['shader_clock_mhz_under_conv_load'] = clock_mhz
                    roof['split_pipe_issue_ceiling_at_held_clock'] = held_b
                    roof['peak_at_held_clock'] = held
                    roof['frac_of_peak_at_held_clock'] = ach / held
            if dom in fam_t and timed_async and not args.no_detail:      # the same family as the timed region saw it (overlapped by the side stream)
                f2 = fam_t[dom]
                roof['timed_region_overlapped'] = {'achieved': f2[0] / f2[1] / 1e12, 'frac': f2[0] / f2[1] / 1e12 / peak, 'avg_launch_ms': f2[1] / f2[2] * 1e3,
                                                   'note': 'event pairs of the timed region: kernels of the other streams share the chip with these launches'}
        roof_hbm = None
        if 'norm_act' in fam_d:      # the largest HBM-bound family: BatchNorm / InstanceNorm / activations / residual adds
            flops, secs, n, nbytes = fam_d['norm_act']
            gbs = nbytes / secs / 1e9
            hb = hbm_per_step('bn_')
            roof_hbm = {'bound': 'hbm', 'kernel': 'norm_act', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                        'traffic': (hb / (n / detail_steps)) if hb is not None else None, 'algorithmic_bytes_per_launch': nbytes / n,
                        'launches': n, 'avg_launch_ms': secs / n * 1e3, 'ms_per_step': secs / detail_steps * 1e3,
                        'note': 'a launch = one normalisation+activation op (forward: statistics unless the conv epilogue made them + apply; '
                                'backward: reduce + apply), timed in %d extra untimed steps' % detail_steps}
        pixels = args.height * args.width
        executed = sum(v[0] for k, v in fam.items()) / prof_steps       # FLOPs the kernels really ran per step (GEMM parts; compat mode runs 2 of the 16 attention calls)
        line = {
            'metric': 'train samples/sec, %s %dx%d DP pair' % ({'psmnet': 'PSMNet', 'nnet': 'NNet', 'stereonet': 'StereoNet'}.get(args.model, 'StereoDPNet'), args.height, args.width), 'value': value, 'unit': 'samples/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': {'f32': 'f32', 'bf16': 'bf16 conv operands (2-D and 3-D), f32 accumulate / tensors / everything else',
                                                      'bf16-2d': 'bf16 2-D conv operands, f32 elsewhere'}[args.precision],
            'data': 'synthetic',
            'config': {'workload': '%s train step (fwd+loss+bwd+grad all-reduce+Adam), %d x %dx%d synthetic DP pairs per GPU'
                                   % ({'psmnet': 'PSMNet', 'nnet': 'NNet', 'stereonet': 'StereoNet'}.get(args.model, 'StereoDPNet'), args.batch, args.height, args.width),
                       'global_batch': global_batch, 'height': args.height, 'width': args.width, 'parallelism': 'dp%d' % world,
                       'batchnorm': 'global-batch statistics (SyncBatchNorm)' if (args.sync_bn and world > 1) else 'per-rank statistics',
                       'streams': ('weight gradients on a side stream, left / right feature passes on two streams' if timed_async else 'one stream'),
                       'launch': ('one HIP graph per step (captured in the warm-up, replayed in the timed region)' if graph_live else 'eager kernel launches')},
            'final_loss': loss,
            'f32_products': {'default': {2: 'three f16 partial products of two-way operand splits, scaled and RANGE-GUARDED per position / output row / channel / voxel (dpf_set_f32_matrix_path(2))',
                                         1: 'six bf16 partial products of exact three-way operand splits (dpf_set_f32_matrix_path(1))',
                                         0: 'v_mfma_f32_* (dpf_set_f32_matrix_path(0))'}.get(mpath_default),
                             'value_with_exact_splits_path1': value_path1,
                             'note': '`value` is measured on the default; the second number is the same step (5 steps, graph re-captured) with every '
                                     'product formed from exact three-way bf16 splits.  Since round 6 every kernel of the default path guards the '
                                     'range of its two-component splits along the axis its outputs do not sum over, so an output element is as accurate '
                                     'relative to its own inputs as on v_mfma_f32_32x32x2_f32 (DESIGN.md section 4; tests/test_gpu_ops.py '
                                     '*_in_block_dynamic_range; tests/test_gpu_e2e.py::test_headline_config_whole_train_step ties it to path 0 at this size)'},
            'rccl_ranks_seen': ranks_seen,
            'collective_backend': (dist.get_backend() if dist.is_initialized() else None),
            'gradient_collectives_per_step': (reducer.collective_calls / float(args.warmup + args.steps + (0 if args.no_detail else 4))) if reducer is not None else 0,
            # whole-model fractions of the fp32 peak: against the reference's algorithmic FLOPs (SURVEY section 8d counts all 16 attention
            # calls) and against the FLOPs the kernels executed (compat mode computes 2 of the 16: identical results)
            'flop_frac_of_f32_peak': (value * FLOP_PER_PIXEL_FWD_BWD * pixels / (world * PEAK_F32_TFLOPS * 1e12)) if args.model == 'stereodpnet' else None,
            'flop_frac_executed': executed * args.steps / elapsed / (PEAK_F32_TFLOPS * 1e12),
            'executed_tflop_per_step': executed / 1e12,
            'roofline': roof,
            'roofline_hbm': roof_hbm,
        }
        if world == 1 and not args.no_cpu_baseline:
            # 16 threads: the oracle's small-tensor PyTorch-CPU ops scale badly beyond that (256 threads on the 128-core
            # host took 740 s for the same sample that 8-16 threads finish in ~10-20 s)
            line['cpu_baseline'] = cpu_baseline(min(16, os.cpu_count() or 1))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
