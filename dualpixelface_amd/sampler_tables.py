"""Host-side sampler tables for the row-shift triple (nearest / bilinear / phase).

The tables reproduce, in the reference's own float32 operation order, where ``subpixel_shift.forward``
(src/module/asm/asm.py:21-127) samples each output pixel:
  * the grid is normalised with the align_corners=True formula ``x/(w-1)*2-1`` (asm.py:40-41);
  * the nearest branch un-normalises with grid_sample's align_corners=False rule ``((g+1)*size-1)/2`` and rounds
    half-to-even (asm.py:96) -- which zero-fills the last column and one or two rows;
  * the bilinear branch un-normalises with ``((g+1)/2)*(size-1)`` (asm.py:101-102): two taps per axis with the
    float32 weights grid_sample uses, zero padding;
  * the Fourier-phase branch multiplies the row spectrum by exp(2*pi*i*delta*k/h) (asm.py:59-75,112-125): for an
    integer ``delta`` that is a circular row roll, out[y] = src[(y + delta) mod h].
Every mode is expressed as <= 2 row taps x <= 2 column taps: iy/wy [3][2][h], ix/wx [3][2][w] (index -1 = no tap).
"""
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


def build_shift_tables(h, w, delta, use_nearest=True, use_bilinear=True, use_phase=True):
    """-> (iy int32 [3,2,h], wy float32 [3,2,h], ix int32 [3,2,w], wx float32 [3,2,w]) on the CPU."""
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
    if float(delta) != float(int(delta)):
        raise NotImplementedError('fractional Fourier-phase shifts (asm_grid_cache_compat=false) are not implemented yet')
    iy[2, 0] = ((torch.arange(h) + int(delta)) % h).to(torch.int32)
    wy[2, 0] = 1.0
    ix[2, 0] = torch.arange(w, dtype=torch.int32)
    wx[2, 0] = 1.0
    return iy.contiguous(), wy.contiguous(), ix.contiguous(), wx.contiguous()


def apply_tables_reference(fea, tables):
    """Slow torch evaluation of the table sampler (used by the CPU tests to pin the host logic against the golden
    vectors; the product path evaluates the same tables in dpf_shift_triple_forward)."""
    iy, wy, ix, wx = tables
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
