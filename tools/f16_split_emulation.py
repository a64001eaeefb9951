"""CPU emulation of the f16-component fp32 product construction (dpf_set_f32_matrix_path(2); DESIGN.md section 4, profiles/r05_f16_split_emulation.txt):
x * 2^s = hi + lo (both f16, round to nearest), three partial products lo*hi, hi*lo, hi*hi summed in fp32 per 16-element block, against fp64 and
against a sequential fp32 FMA chain.  Rows of the test matrix are scaled by `dyn` to show what the BLOCK scaling costs values far below the
block's maximum (the scale of a block is set by its largest magnitude).      python tools/f16_split_emulation.py"""
import numpy as np

rng = np.random.default_rng(0)
n, rows = 864, 4000


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def split2(x, s):
    xs = (x * np.float32(s)).astype(np.float32)
    h = f16(xs)
    return h, f16((xs - h).astype(np.float32))


print('864-term dot products (32 channels x 27 taps), 4000 rows; relative error of each row against fp64: median | max over the rows')
print('%-10s %-8s | %-26s | %-26s | %-26s | %s' % ('data', 'dyn', 'fp32 FMA chain', 'f16 x3, rows at block max', 'f16 x3, rows dyn below max', 'products only (max)'))
for positive in (False, True):
    for dyn in (1.0, 2.0 ** -13, 2.0 ** -20, 2.0 ** -27):
        x = rng.standard_normal((rows, n)).astype(np.float32)
        w = (0.1 * rng.standard_normal((rows, n))).astype(np.float32)
        if positive:
            x, w = np.abs(x), np.abs(w)
        half = rows // 2
        x[:half] *= np.float32(dyn)                     # the first half of the rows lies `dyn` below the block maximum
        ref = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
        acc = np.zeros(rows, np.float32)
        for i in range(n):
            acc = (acc.astype(np.float64) + x[:, i].astype(np.float64) * w[:, i]).astype(np.float32)
        e32 = np.abs(acc - ref) / np.abs(ref)
        sx = 2.0 ** (14 - np.floor(np.log2(np.abs(x).max())))
        sw = 2.0 ** (14 - np.floor(np.log2(np.abs(w).max())))
        xh, xl = split2(x, sx)
        wh, wl = split2(w, sw)
        assert np.isfinite(xh).all() and np.isfinite(wh).all()
        acc = np.zeros(rows, np.float32)
        for blk in range(0, n, 16):
            for a, b in ((wl, xh), (wh, xl), (wh, xh)):
                part = (a[:, blk:blk + 16].astype(np.float64) * b[:, blk:blk + 16]).sum(1)
                acc = (acc.astype(np.float64) + part).astype(np.float32)
        e16 = np.abs(acc.astype(np.float64) / (sx * sw) - ref) / np.abs(ref)
        tot = sum((a.astype(np.float64) * b).sum(1) for a, b in ((wl, xh), (wh, xl), (wh, xh))) / (sx * sw)
        ep = np.abs(tot - ref) / np.abs(ref)
        print('%-10s 2^%-6d | %.2e | %.2e       | %.2e | %.2e       | %.2e | %.2e       | big %.1e small %.1e'
              % ('positive' if positive else 'signed', int(np.log2(dyn)), np.median(e32), e32.max(), np.median(e16[half:]), e16[half:].max(),
                 np.median(e16[:half]), e16[:half].max(), ep[half:].max(), ep[:half].max()))
