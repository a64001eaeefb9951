"""ORACLE (test infrastructure only) -- PSMNet's integer-shift cost volume
(/root/reference/src/model/psmnet/modules.py:215-275), pinned by tests/golden/psmnet_volume.npz."""
import torch


def psm_volume(ref, tar, costrange, groups=0):
    B, C, H, W = ref.shape
    L = len(costrange)
    vol = ref.new_zeros(B, 2 * C + groups, L, H, W)
    for i, disp in enumerate(costrange):
        d = int(disp)                                   # truncation toward zero (:229)
        if d == 0:
            r, t, rows = ref, tar, slice(None)
        elif d > 0:
            r, t, rows = ref[:, :, :-d], tar[:, :, d:], slice(0, H - d)
        else:
            r, t, rows = ref[:, :, -d:], tar[:, :, :d], slice(-d, H)
        vol[:, :C, i, rows] = r
        vol[:, C:2 * C, i, rows] = t
        if groups:
            corr = (r * t).view(B, groups, C // groups, r.shape[2], W).mean(2)
            vol[:, 2 * C:, i, rows] = -corr
    return vol
