"""CPU restatement of the f16-component product construction behind dpf_set_f32_matrix_path(2) (conv_internal.h: dpf_split_pair_h, the block
scaling of igemm3_x9_kernel / wgrad2_kernel): numpy float16 rounds to nearest even like v_cvt_pk_f16_f32.  The GPU counterparts are
tests/test_gpu_ops.py::test_conv_f32_matrix_paths_agree, test_weight_gradient_f32_matrix_paths_agree and
test_conv_f16_component_path_block_scaling; the full table is profiles/r05_f16_split_emulation.txt (tools/f16_split_emulation.py)."""
import numpy as np


def f16(x):
    return x.astype(np.float16).astype(np.float32)


def split(x, e):
    """x * 2^(141 - E) = hi + lo + err, E = biased exponent of the block maximum (clamped like DPF_H3_EMIN .. 254)."""
    s = np.float32(2.0) ** np.float32(141 - e)
    xs = (x * s).astype(np.float32)
    hi = f16(xs)
    lo = f16((xs - hi).astype(np.float32))
    return hi, lo, s


def block_exponent(x):
    m = np.abs(x).max()
    e = int(np.float32(m).view(np.uint32) >> 23) if m > 0 else 0
    return min(max(e, 14), 254)


def dot3(x, w):
    """three partial products, summed in fp32 per 16-element block like the MFMA's accumulator"""
    ex, ew = block_exponent(x), block_exponent(w)
    xh, xl, sx = split(x, ex)
    wh, wl, sw = split(w, ew)
    assert np.isfinite(xh).all() and np.isfinite(wh).all() and np.abs(xh).max() <= 32768 and np.abs(wh).max() <= 32768
    acc = np.zeros(x.shape[0], np.float32)
    for b in range(0, x.shape[1], 16):
        for a, c in ((wl, xh), (wh, xl), (wh, xh)):
            acc = (acc.astype(np.float64) + (a[:, b:b + 16].astype(np.float64) * c[:, b:b + 16]).sum(1)).astype(np.float32)
    return np.ldexp(acc.astype(np.float64), ex + ew - 282)


def fma_chain(x, w):
    acc = np.zeros(x.shape[0], np.float32)
    for i in range(x.shape[1]):
        acc = (acc.astype(np.float64) + x[:, i].astype(np.float64) * w[:, i]).astype(np.float32)
    return acc.astype(np.float64)


def test_components_represent_an_operand_to_one_ulp():
    rng = np.random.default_rng(1)
    x = (rng.choice([-1.0, 1.0], 20000) * (1.0 + rng.random(20000)) * np.exp2(rng.integers(-16, 1, 20000))).astype(np.float32)   # within 2^17 of the block maximum
    hi, lo, s = split(x, block_exponent(x))
    err = np.abs((hi.astype(np.float64) + lo) / s - x) / np.abs(x)
    assert err.max() <= 2.0 ** -22 and np.median(err) <= 2.0 ** -25
    assert np.abs(lo).max() <= np.abs(hi).max() * 2.0 ** -10


def test_three_products_are_as_accurate_as_an_fp32_chain():
    rng = np.random.default_rng(2)
    for positive in (False, True):
        x = rng.standard_normal((300, 864)).astype(np.float32)
        w = (0.1 * rng.standard_normal((300, 864))).astype(np.float32)
        if positive:
            x, w = np.abs(x), np.abs(w)
        ref = (x.astype(np.float64) * w).sum(1)
        scale = np.abs(ref).max()
        e3, e32 = np.abs(dot3(x, w) - ref).max() / scale, np.abs(fma_chain(x, w) - ref).max() / scale
        assert e3 <= 1e-6 and e3 <= 2 * e32 + 1e-8, (positive, e3, e32)


def test_power_of_two_scaling_is_exact_and_rows_far_below_the_block_maximum_degrade_gracefully():
    rng = np.random.default_rng(3)
    x = np.abs(rng.standard_normal((64, 864))).astype(np.float32)
    w = np.abs(0.1 * rng.standard_normal((64, 864))).astype(np.float32)
    base = dot3(x, w)
    for k, j in ((-100, 0), (90, -60), (-50, -50)):
        assert np.array_equal(dot3(x * np.float32(2.0) ** k, w * np.float32(2.0) ** j), base * 2.0 ** (k + j))
    # half of the rows 2^20 / 2^27 below the block maximum: 2^-23 .. 2^-13 relative, never garbage; the rows at the maximum are unaffected
    for shift, bound in ((-20, 4e-6), (-27, 5e-4)):
        y = x.copy()
        y[:32] *= np.float32(2.0) ** shift
        ref = (y.astype(np.float64) * w).sum(1)
        rel = np.abs(dot3(y, w) - ref) / np.abs(ref)
        assert rel[:32].max() <= bound and rel[32:].max() <= 1e-6, (shift, rel[:32].max(), rel[32:].max())
    assert np.array_equal(dot3(np.zeros_like(x), w), np.zeros(64))
