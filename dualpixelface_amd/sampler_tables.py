"""Host-side sampler tables for the row-shift triple (nearest / bilinear / phase).

The tables reproduce, in the reference's own float32 operation order, where ``subpixel_shift.forward``
(src/module/asm/asm.py:21-127) samples each output pixel:
  * the grid is normalised with the align_corners=True formula ``x/(w-1)*2-1`` (asm.py:40-41);
  * the nearest branch un-normalises with grid_sample's align_corners=False rule ``((g+1)*size-1)/2`` and rounds
    half-to-even (asm.py:96) -- which zero-fills the last column and one or two rows;
  * the bilinear branch un-normalises with ``((g+1)/2)*(size-1)`` (asm.py:101-102): two taps per axis with the
    float32 weights grid_sample uses, zero padding;
  * the Fourier-phase branch multiplies the row spectrum by exp(2*pi*i*delta*k/h) (asm.py:59-75,112-125): for an
    integer ``delta`` that is a circular row roll, out[y] = src[(y + delta) mod h]; for a fractional ``delta`` (only reached
    with per-level shifts, ``asm_grid_cache_compat = false``) it is a dense row-circulant plus a rank-one Hilbert term
    (``build_phase_tables``), evaluated by dpf_phase_shift into slot 2 of the triple.
Every table mode is expressed as <= 2 row taps x <= 2 column taps: iy/wy [3][2][h], ix/wx [3][2][w] (index -1 = no tap); the
inverse tables iy_inv / ix_inv (source coordinate -> output coordinate, the taps are injective) drive the deterministic adjoint.
"""
import math

import numpy as np
import torch


def _axis_tables(n, delta):
    """nearest and bilinear taps along one axis of length n for a shift ``delta`` (float32 throughout)."""
    dt = torch.float32
    c = torch.arange(0.0, n, dtype=dt) + torch.tensor(float(delta), dtype=dt)
    g = c / (n - 1) * 2.0 - 1.0
    # nearest, align_corners=False un-normalisation
    u = ((g + 1.0) * n - 1.0) / 2.0
    r = torch.round(u)                                   # std::nearbyint: half to even
    near_i = torch.where((r >= 0) & (r <= n - 1), r, torch.full_like(r, -1.0)).to(torch.int32)
    # bilinear, align_corners=True un-normalisation
    v = ((g + 1.0) / 2.0) * (n - 1)
    lo = torch.floor(v)
    hi = lo + 1.0
    w_lo = hi - v
    w_hi = v - lo
    i_lo = torch.where((lo >= 0) & (lo <= n - 1), lo, torch.full_like(lo, -1.0)).to(torch.int32)
    i_hi = torch.where((hi >= 0) & (hi <= n - 1), hi, torch.full_like(hi, -1.0)).to(torch.int32)
    return near_i, (i_lo, i_hi, w_lo, w_hi)


def is_fractional(delta):
    return float(delta) != float(int(delta))


INV_SLOTS = 2


def _invert(idx, n):
    """idx [3,2,n] (output coordinate -> source coordinate, -1 = none) -> inverse [3,2,INV_SLOTS,n]: the (at most INV_SLOTS) output
    coordinates that read a given source coordinate through tap (m, a), -1 padded.  The maps are monotone; float32 rounding of the
    bilinear coordinate can make two neighbouring outputs share a source row, never more."""
    inv = torch.full((idx.shape[0], idx.shape[1], INV_SLOTS, n), -1, dtype=idx.dtype)
    for m in range(idx.shape[0]):
        for a in range(idx.shape[1]):
            fill = [0] * n
            for out_pos, src_pos in enumerate(idx[m, a].tolist()):
                if src_pos < 0:
                    continue
                if fill[src_pos] >= INV_SLOTS:
                    raise RuntimeError('sampler tap map (%d, %d): more than %d outputs read source %d' % (m, a, INV_SLOTS, src_pos))
                inv[m, a, fill[src_pos], src_pos] = out_pos
                fill[src_pos] += 1
    return inv.contiguous()


def build_phase_tables(h, w, delta):
    """Fractional Fourier-phase shift as out = mr (*)_rows src + scale (-1)^y (hm (*)_cols S), S = alternating column sums (see
    csrc/costvolume.hip).  -> (mr [h], hm [w], scale, mr_T [h], hm_T [w]) fp32; the *_T tables give the adjoint."""
    if h % 2:
        raise NotImplementedError('the Fourier-phase shift needs an even feature height (asm.py:68-69 fails otherwise too)')
    nr = np.concatenate([np.arange(0, h // 2), np.arange(-(h // 2), 0)]).astype(np.float64)
    M = np.exp(1j * 2.0 * np.pi * (float(delta) / h) * nr)
    Mh = M.copy()
    Mh[h // 2] = M[h // 2].real                                    # the Hermitian part of the Nyquist bin
    mr = np.fft.ifft(Mh).real
    scale = -M[h // 2].imag / h                                    # sin(pi delta) / h
    d = np.arange(w)
    k = np.arange(1, (w - 1) // 2 + 1)
    hm = (2.0 / w) * np.sin(2.0 * np.pi * np.outer(d, k) / w).sum(1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    rev = lambda a: np.concatenate([a[:1], a[:0:-1]])               # a[(-i) mod n]
    return t(mr), t(hm), float(scale), t(rev(mr)), t(rev(hm))


def build_shift_tables(h, w, delta, use_nearest=True, use_bilinear=True, use_phase=True):
    """-> (iy int32 [3,2,h], wy float32 [3,2,h], ix int32 [3,2,w], wx float32 [3,2,w], iy_inv, ix_inv) on the CPU.  For a fractional
    delta the phase slot (mode 2) has no taps: it is filled by dpf_phase_shift (build_phase_tables)."""
    if not (use_nearest and use_bilinear and use_phase):
        raise NotImplementedError('the HIP cost-volume path implements the shipped nearest+bilinear+phase triple')
    iy = torch.full((3, 2, h), -1, dtype=torch.int32)
    ix = torch.full((3, 2, w), -1, dtype=torch.int32)
    wy = torch.zeros((3, 2, h), dtype=torch.float32)
    wx = torch.zeros((3, 2, w), dtype=torch.float32)
    ny, (ylo, yhi, wylo, wyhi) = _axis_tables(h, delta)
    nx, (xlo, xhi, wxlo, wxhi) = _axis_tables(w, 0.0)
    # mode 0: nearest
    iy[0, 0], wy[0, 0] = ny, 1.0
    ix[0, 0], wx[0, 0] = nx, 1.0
    # mode 1: bilinear
    iy[1, 0], iy[1, 1], wy[1, 0], wy[1, 1] = ylo, yhi, wylo, wyhi
    ix[1, 0], ix[1, 1], wx[1, 0], wx[1, 1] = xlo, xhi, wxlo, wxhi
    # mode 2: phase shift == circular roll for an integer delta
    if not is_fractional(delta):
        iy[2, 0] = ((torch.arange(h) + int(delta)) % h).to(torch.int32)
        wy[2, 0] = 1.0
        ix[2, 0] = torch.arange(w, dtype=torch.int32)
        wx[2, 0] = 1.0
    iy, ix = iy.contiguous(), ix.contiguous()
    return iy, wy.contiguous(), ix, wx.contiguous(), _invert(iy, h), _invert(ix, w)


def apply_tables_reference(fea, tables):
    """Slow torch evaluation of the table sampler (used by the CPU tests to pin the host logic against the golden
    vectors; the product path evaluates the same tables in dpf_shift_triple_forward)."""
    iy, wy, ix, wx = tables[:4]
    B, C, h, w = fea.shape
    out = fea.new_zeros(B, C, 3, h, w)
    for m in range(3):
        for a in range(2):
            for e in range(2):
                ry, rx = iy[m, a].long(), ix[m, e].long()
                ok = ((ry >= 0).view(h, 1) & (rx >= 0).view(1, w)).to(fea.dtype)
                wgt = (wy[m, a].view(h, 1) * wx[m, e].view(1, w)) * ok
                out[:, :, m] += wgt * fea[:, :, ry.clamp(min=0)][:, :, :, rx.clamp(min=0)]
    return out
