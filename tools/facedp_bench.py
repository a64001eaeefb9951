#!/usr/bin/env python3
"""Throughput of the FaceDP data path on the GPU box, full-size 1024 x 1536 frames, shipped training
augmentation (random 768 x 1152 crop + lighting, use_raw):

  kernels    the device preprocessing of one sample (HIP events): algorithmic bytes / time vs the HBM roof
  batcher    FaceDPBatcher end to end from files on disk (JPEG decode threads + upload + kernels), samples/s
  cpu        the numpy oracle doing the same per-sample arithmetic on one host core (the reference's DataLoader-worker work)

Writes one JSON line.  Usage: python tools/facedp_bench.py [--samples 16] [--workers 8] [--out gpurun_out/facedp_bench.json]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dualpixelface_amd import facedp, load_option  # noqa: E402

H, W = 1024, 1536


def build_dataset(root, samples, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(root)
    with open(os.path.join(root, 'train.txt'), 'w') as fh:
        fh.write('subj\n')
    base = os.path.join(root, 'subj')
    for sub in ('JSON', 'IMG/LEFT', 'IMG/RIGHT', 'IMG/LRSUM', 'DEPTH', 'NORMAL'):
        os.makedirs(os.path.join(base, sub))
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(samples):
        rad = np.hypot((yy - H / 2) / (0.45 * H), (xx - W / 2) / (0.45 * W))
        face = rad < 1
        np.save(os.path.join(base, 'DEPTH', 'DEPTH_1_%d.npy' % i), np.where(face, 950 + 100 * rad ** 2, 0).astype(np.float32))
        normal = rng.normal(size=(H, W, 3)).astype(np.float32)
        np.save(os.path.join(base, 'NORMAL', 'NORMAL_1_%d.npy' % i), normal * face[..., None])
        smooth = (127 + 100 * np.sin(xx / 37.0 + i) * np.cos(yy / 23.0)).astype(np.int32)
        for side in ('LEFT', 'RIGHT', 'LRSUM'):
            img = np.clip(smooth[..., None] + rng.randint(-20, 20, (H, W, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(base, 'IMG', side, 'IMG_1_%d_1.JPG' % i), quality=92)
        stem = 'IMG_1_%d_1.JPG' % i
        info = {'valid': True, 'object': 'subj', 'gender': 'w', 'camidx': 1, 'lightidx': 1, 'expression': 'neutral', 'position': 'forward',
                'direction': 'front'}
        paths = {'left': 'IMG/LEFT/' + stem, 'right': 'IMG/RIGHT/' + stem, 'lrsum': 'IMG/LRSUM/' + stem,
                 'depth': 'DEPTH/DEPTH_1_%d.npy' % i, 'normal': 'NORMAL/NORMAL_1_%d.npy' % i}
        params = {'intrinsic': repr(np.array([7000., 7000., 0., W / 2, H / 2, 0, 0, 0, 0])), 'pose': repr(np.arange(12.)), 'Lvalue': None}
        with open(os.path.join(base, 'JSON', 'INFO_1_%d_1.json' % i), 'w') as fh:
            json.dump({'INFO': info, 'PATH': paths, 'PARAMS': params}, fh)
    return root


def sample_bytes(raw, plan, use_raw):
    """Algorithmic HBM bytes of one sample's kernels: every source window read once, every output written once."""
    h, w = raw.image_shape()
    n_img = sum(getattr(raw, n) is not None for n in ('left', 'right', 'center'))
    def view(ch, cw):
        px = ch * cw
        return n_img * px * (3 + 12) + px * (4 + 16) + (px * 24 if raw.normal is not None else 0)
    total = h * w * 4                                               # stats pass over the depth
    total += view(plan.ch, plan.cw)
    if use_raw:
        total += view(h, w)
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=16)
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--cpu-samples', type=int, default=3)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix='facedp_bench_')
    data = build_dataset(os.path.join(tmp, 'data'), args.samples)
    os.chdir(tmp)
    opt = load_option('train_faceDP', root=ROOT)
    opt.dataset.path = data
    opt.dataset.viewpoint = [1]
    ds = facedp.FaceDPLoader(opt, True, device='cuda:0')
    pre = ds.preprocessor()

    # ---- kernels only
    item = ds.read(0)
    raw, flags, _ = item
    plan = facedp.draw_plan(opt, raw, flags)
    dev = pre.upload(raw)
    win = (plan.x0, plan.y0, plan.ch, plan.cw)
    out = facedp._allocate_view(dev, flags, plan.ch, plan.cw, pre.device)
    rout = facedp._allocate_view(dev, flags, H, W, pre.device, 'raw_')
    lib_call = pre.lib.call

    def kernels():
        d = dev['depth']
        lib_call('dpf_dp_depth_stats', facedp._ptr(d), 0, None, d.numel(), raw.a, raw.b, facedp._ptr(dev['stats']), pre._stream())
        facedp._fill_view(pre, dev, raw, flags, win, out, '', plan=plan, keep=[])
        facedp._fill_view(pre, dev, raw, flags, (0, 0, H, W), rout, 'raw_', plan=None)
    for _ in range(5):
        kernels()
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        kernels()
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / reps
    nbytes = sample_bytes(raw, plan, True)

    # ---- batcher end to end (files -> device batches)
    def run_batcher(workers):
        b = facedp.FaceDPBatcher(ds, batch_size=4, shuffle=True, workers=workers)
        n = 0
        t0 = time.time()
        for epoch in range(2):
            b.set_epoch(epoch)
            for batch in b:
                n += batch['left'].shape[0]
        torch.cuda.synchronize()
        return n / (time.time() - t0)
    run_batcher(args.workers)                                      # page cache warm
    rates = {w: run_batcher(w) for w in sorted({1, 4, args.workers})}

    # ---- host-only stage timings of one sample
    t0 = time.time()
    for i in range(4):
        ds.read(i)
    read_ms = (time.time() - t0) / 4 * 1e3
    t0 = time.time()
    for _ in range(4):
        d2 = pre.upload(raw)
    torch.cuda.synchronize()
    upload_ms = (time.time() - t0) / 4 * 1e3

    # ---- CPU: the oracle's restatement of the same arithmetic, one core
    from oracle import facedp_preprocess as oracle
    torch.set_num_threads(1)
    arrays = {n: getattr(raw, n) for n in ('left', 'right', 'center', 'depth', 'file_mask', 'normal', 'albedo')}
    shifts = {k: (v.numpy() if v is not None else None) for k, v in plan.shift.items()}
    t0 = time.time()
    for _ in range(args.cpu_samples):
        oracle.sample_view(arrays, flags, win, shifts, None, True, raw.a, raw.b)
        oracle.sample_view(arrays, flags, (0, 0, H, W), None, None, False, raw.a, raw.b)
    cpu_ms = (time.time() - t0) / args.cpu_samples * 1e3

    rec = {
        'workload': 'FaceDP 1024x1536 sample -> 768x1152 crop + raw views (config_train augmentation, use_raw)',
        'kernel_ms_per_sample': round(k_ms, 4), 'algorithmic_MB_per_sample': round(nbytes / 1e6, 2),
        'kernel_GBps': round(nbytes / k_ms / 1e6, 1), 'hbm_frac': round(nbytes / k_ms / 1e6 / 8000.0, 4),
        'batcher_samples_per_s': {str(k): round(v, 2) for k, v in rates.items()},
        'host_read_decode_ms_per_sample_1thread': round(read_ms, 2), 'upload_ms_per_sample': round(upload_ms, 2),
        'cpu_oracle_ms_per_sample_1core': round(cpu_ms, 1), 'cpu_cores': os.cpu_count(),
    }
    line = json.dumps(rec)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(os.path.join(ROOT, args.out))), exist_ok=True)
        with open(os.path.join(ROOT, args.out), 'w') as fh:
            fh.write(line + '\n')


if __name__ == '__main__':
    main()
