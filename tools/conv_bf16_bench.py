"""Per-shape timing of the dense conv kernels with fp32 vs bf16 operands (ops.conv_operands).  usage: python tools/conv_bf16_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualpixelface_amd import ops  # noqa: E402

SHAPES = [
    # N, C, K, D, H, W, k(3), stride, pad(3), dil
    (4, 32, 32, 8, 256, 384, (3, 3, 3), 1, (1, 1, 1), 1),
    (4, 64, 81, 4, 256, 384, (3, 3, 3), 1, (1, 1, 1), 1),
    (4, 64, 32, 8, 256, 384, (3, 3, 3), 1, (1, 1, 1), 1),
    (4, 32, 64, 8, 256, 384, (3, 3, 3), 2, (1, 1, 1), 1),
    (4, 64, 64, 4, 128, 192, (3, 3, 3), 1, (1, 1, 1), 1),
    (4, 32, 32, 1, 512, 768, (1, 3, 3), 1, (0, 1, 1), 1),
    (16, 96, 96, 1, 256, 384, (1, 3, 3), 1, (0, 2, 2), 2),
    (16, 64, 64, 1, 256, 384, (1, 3, 3), 1, (0, 8, 8), 8),
    (4, 64, 64, 1, 128, 192, (1, 3, 3), 1, (0, 1, 1), 1),
    (4, 32, 32, 1, 256, 384, (1, 3, 3), 1, (0, 1, 1), 1),
]
REPS = 10


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


def main():
    dev = 'cuda:0'
    for N, C, K, D, H, W, ks, st, pd, dl in SHAPES:
        x = torch.randn(N, C, D, H, W, device=dev)
        w = torch.randn(K, C, *ks, device=dev) * 0.05
        s3, d3 = (st if ks[0] > 1 else 1, st, st), (dl if ks[0] > 1 else 1, dl, dl)
        y = ops._conv_fwd_raw(x, w, None, s3, pd, d3)
        g = torch.randn_like(y)
        flops = 2.0 * N * K * C * ks[0] * ks[1] * ks[2] * y.shape[2] * y.shape[3] * y.shape[4]
        row = '%-44s' % ('N%d C%d K%d %dx%dx%d k%d%d%d s%d d%d' % (N, C, K, D, H, W, *ks, st, dl))
        for name, fn in (('fwd', lambda: ops._conv_fwd_raw(x, w, None, s3, pd, d3)),
                         ('dgrad', lambda: ops._conv_transpose_raw(g, w, None, x.shape[2:], ks, s3, pd, d3)),
                         ('wgrad', lambda: ops._conv_wgrad_raw(g, x, w.shape, s3, pd, d3))):
            t32 = timed(fn)
            with ops.conv_operands(True):
                t16 = timed(fn)
            row += '  %s %6.3f -> %6.3f ms (%5.1f -> %6.1f TF)' % (name, t32, t16, flops / t32 / 1e9, flops / t16 / 1e9)
        print(row, flush=True)


if __name__ == '__main__':
    main()
