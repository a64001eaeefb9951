"""CPU emulation of the bf16-split fp32 product constructions (DESIGN.md section 4 "Round 5"; profiles/r05_x6_split_emulation.txt):
fp32 FMA chain vs truncation / round-to-nearest three-way splits with 9 / 8 / 6 / 4 partial products, against fp64.   python tools/x6_split_emulation.py"""
import numpy as np, torch
def bf16_rn(x):  # x float32 np -> float32 rounded to bf16 RNE
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
def bf16_tr(x):
    u = x.view(np.uint32) & np.uint32(0xffff0000)
    return u.view(np.float32)
def split(x, rn):
    f = bf16_rn if rn else bf16_tr
    h = f(x); r = (x - h).astype(np.float32); m = f(r); l = (r - m).astype(np.float32)
    assert np.all(f(l) == l)
    return h, m, l
rng = np.random.default_rng(0)
n, rows = 864, 4000
for positive in (False, True):
    x = rng.standard_normal((rows, n)).astype(np.float32); w = (0.1 * rng.standard_normal((rows, n))).astype(np.float32)
    if positive: x, w = np.abs(x), np.abs(w)
    ref = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
    # fp32 fma chain emulation: sequential fp32 accumulate of exact products (approx: products in f64, acc rounded to f32 each step)
    acc = np.zeros(rows, np.float32)
    for i in range(n):
        acc = (acc.astype(np.float64) + x[:, i].astype(np.float64) * w[:, i]).astype(np.float32)
    e32 = np.abs(acc - ref).max() / np.abs(ref).max()
    out = {}
    for rn in (False, True):
        xs, ws = split(x, rn), split(w, rn)
        for first, name in ((1, 'x8'), (3, 'x6')):
            oa = [2, 2, 1, 2, 0, 1, 1, 0, 0]; ob = [2, 1, 2, 0, 2, 1, 0, 1, 0]
            # accumulate per 16-element block: each MFMA adds sum over 16 products (exact-ish inside: emulate with f64 sum then f32 round)
            acc = np.zeros(rows, np.float32)
            for blk in range(0, n, 16):
                for i in range(first, 9):
                    part = (ws[oa[i]][:, blk:blk+16].astype(np.float64) * xs[ob[i]][:, blk:blk+16]).sum(1)
                    acc = (acc.astype(np.float64) + part).astype(np.float32)
            out[(rn, name)] = np.abs(acc - ref).max() / np.abs(ref).max()
    print('positive' if positive else 'signed', 'fp32 chain %.2e' % e32, {k: '%.2e' % v for k, v in out.items()})
print('--- signed mean relative error on positive data (bias) and rms')
x = np.abs(rng.standard_normal((rows, n))).astype(np.float32); w = np.abs(0.1 * rng.standard_normal((rows, n))).astype(np.float32)
ref = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
for rn in (False, True):
    xs, ws = split(x, rn), split(w, rn)
    for first in (0, 1, 3, 5):
        oa = [2, 2, 1, 2, 0, 1, 1, 0, 0]; ob = [2, 1, 2, 0, 2, 1, 0, 1, 0]
        tot = np.zeros(rows, np.float64)
        for i in range(first, 9):
            tot += (ws[oa[i]].astype(np.float64) * xs[ob[i]]).sum(1)
        rel = (tot - ref) / ref
        print('rn' if rn else 'trunc', 'first', first, 'products-only: mean %.3e  max|.| %.3e' % (rel.mean(), np.abs(rel).max()))
