"""ORACLE (test infrastructure only) -- loop-literal scalar restatement of the D3D deformable-conv CUDA kernels.

A second, independent reading of the reference's three kernels, written index by index the way the CUDA threads walk them,
to cross-check the vectorised restatement in ``oracle/dcn3d.py`` on adversarial samples (positions in (-1, 0), exactly on
integers, in (size-1, size), exactly -1 / size, far outside).  It is deliberately slow (pure Python loops, fp64) and only
used on tiny tensors by ``tests/test_oracle_dcn.py``.  Parity status: like ``oracle/dcn3d.py`` it cannot be pinned against
the reference binary (CUDA-only extension, no numeric test upstream) -- two independent restatements agreeing is the
strongest pin available here.

Reference lines followed (all in /root/reference/src/module/dcn3d/src/cuda/deform_im2col_cuda.cuh):
  * trilinear sampler with per-corner bounds tests ........................ :26-72   (``_bilinear``)
  * gradient weight of one input voxel for one sample ...................... :74-109  (``_gradient_weight``)
  * coordinate weight (d sample / d coordinate), 3 directions x 8 corners .. :111-190 (``_coordinate_weight``)
  * deformable_im2col_gpu_kernel ........................................... :192-265 (``im2col``)
  * deformable_col2im_gpu_kernel: ``(int)`` truncation, 5x5x5 window, |p - v| < 1 .. :267-334 (``col2im``)
  * deformable_col2im_coord_gpu_kernel: invalid samples moved to -2 .......... :336-405 (``col2im_coord``)
and deform_conv_cuda.cu:93-123, 226-277 for the GEMMs around them (``forward`` / ``backward``).
Columns are laid out [c * T + tap][b][voxel] as in the reference (cuh:207-211).
"""
import math

import numpy as np


def _dims(x, ksize, stride, pad, dil):
    B, C, D, H, W = x.shape
    kd, kh, kw = ksize
    Do = (D + 2 * pad[0] - (dil[0] * (kd - 1) + 1)) // stride[0] + 1
    Ho = (H + 2 * pad[1] - (dil[1] * (kh - 1) + 1)) // stride[1] + 1
    Wo = (W + 2 * pad[2] - (dil[2] * (kw - 1) + 1)) // stride[2] + 1
    return B, C, D, H, W, Do, Ho, Wo


def _bilinear(im, depth, height, width, d, h, w):                      # cuh:26-72
    d_low, h_low, w_low = math.floor(d), math.floor(h), math.floor(w)
    d_high, h_high, w_high = d_low + 1, h_low + 1, w_low + 1
    ld, lh, lw = d - d_low, h - h_low, w - w_low
    hd, hh, hw = 1 - ld, 1 - lh, 1 - lw
    v1 = im[d_low, h_low, w_low] if (d_low >= 0 and h_low >= 0 and w_low >= 0) else 0.0
    v2 = im[d_low, h_low, w_high] if (d_low >= 0 and h_low >= 0 and w_high <= width - 1) else 0.0
    v3 = im[d_low, h_high, w_low] if (d_low >= 0 and h_high <= height - 1 and w_low >= 0) else 0.0
    v4 = im[d_low, h_high, w_high] if (d_low >= 0 and h_high <= height - 1 and w_high <= width - 1) else 0.0
    v5 = im[d_high, h_low, w_low] if (d_high <= depth - 1 and h_low >= 0 and w_low >= 0) else 0.0
    v6 = im[d_high, h_low, w_high] if (d_high <= depth - 1 and h_low >= 0 and w_high <= width - 1) else 0.0
    v7 = im[d_high, h_high, w_low] if (d_high <= depth - 1 and h_high <= height - 1 and w_low >= 0) else 0.0
    v8 = im[d_high, h_high, w_high] if (d_high <= depth - 1 and h_high <= height - 1 and w_high <= width - 1) else 0.0
    w1, w2, w3, w4 = hd * hh * hw, hd * hh * lw, hd * lh * hw, hd * lh * lw
    w5, w6, w7, w8 = ld * hh * hw, ld * hh * lw, ld * lh * hw, ld * lh * lw
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8


def _gradient_weight(ad, ah, aw, d, h, w, depth, height, width):       # cuh:74-109
    if ad <= -1 or ad >= depth or ah <= -1 or ah >= height or aw <= -1 or aw >= width:
        return 0.0
    dl, hl, wl = math.floor(ad), math.floor(ah), math.floor(aw)
    dh_, hh_, wh_ = dl + 1, hl + 1, wl + 1
    weight = 0.0
    if d == dl and h == hl and w == wl:
        weight = (d + 1 - ad) * (h + 1 - ah) * (w + 1 - aw)
    if d == dl and h == hl and w == wh_:
        weight = (d + 1 - ad) * (h + 1 - ah) * (aw + 1 - w)
    if d == dl and h == hh_ and w == wl:
        weight = (d + 1 - ad) * (ah + 1 - h) * (w + 1 - aw)
    if d == dl and h == hh_ and w == wh_:
        weight = (d + 1 - ad) * (ah + 1 - h) * (aw + 1 - w)
    if d == dh_ and h == hl and w == wl:
        weight = (ad + 1 - d) * (h + 1 - ah) * (w + 1 - aw)
    if d == dh_ and h == hl and w == wh_:
        weight = (ad + 1 - d) * (h + 1 - ah) * (aw + 1 - w)
    if d == dh_ and h == hh_ and w == wl:
        weight = (ad + 1 - d) * (ah + 1 - h) * (w + 1 - aw)
    if d == dh_ and h == hh_ and w == wh_:
        weight = (ad + 1 - d) * (ah + 1 - h) * (aw + 1 - w)
    return weight


def _coordinate_weight(ad, ah, aw, depth, height, width, im, bp_dir):  # cuh:111-190
    if ad <= -1 or ad >= depth or ah <= -1 or ah >= height or aw <= -1 or aw >= width:
        return 0.0
    dl, hl, wl = math.floor(ad), math.floor(ah), math.floor(aw)
    dh_, hh_, wh_ = dl + 1, hl + 1, wl + 1
    weight = 0.0
    lo_d, lo_h, lo_w = dl >= 0, hl >= 0, wl >= 0
    hi_d, hi_h, hi_w = dh_ <= depth - 1, hh_ <= height - 1, wh_ <= width - 1
    if bp_dir == 0:
        if lo_d and lo_h and lo_w:
            weight += -1 * (hl + 1 - ah) * (wl + 1 - aw) * im[dl, hl, wl]
        if lo_d and lo_h and hi_w:
            weight += -1 * (hl + 1 - ah) * (aw - wl) * im[dl, hl, wh_]
        if lo_d and hi_h and lo_w:
            weight += -1 * (ah - hl) * (wl + 1 - aw) * im[dl, hh_, wl]
        if lo_d and hi_h and hi_w:
            weight += -1 * (ah - hl) * (aw - wl) * im[dl, hh_, wh_]
        if hi_d and lo_h and lo_w:
            weight += (hl + 1 - ah) * (wl + 1 - aw) * im[dh_, hl, wl]
        if hi_d and lo_h and hi_w:
            weight += (hl + 1 - ah) * (aw - wl) * im[dh_, hl, wh_]
        if hi_d and hi_h and lo_w:
            weight += (ah - hl) * (wl + 1 - aw) * im[dh_, hh_, wl]
        if hi_d and hi_h and hi_w:
            weight += (ah - hl) * (aw - wl) * im[dh_, hh_, wh_]
    elif bp_dir == 1:
        if lo_d and lo_h and lo_w:
            weight += -1 * (dl + 1 - ad) * (wl + 1 - aw) * im[dl, hl, wl]
        if lo_d and lo_h and hi_w:
            weight += -1 * (dl + 1 - ad) * (aw - wl) * im[dl, hl, wh_]
        if lo_d and hi_h and lo_w:
            weight += (dl + 1 - ad) * (wl + 1 - aw) * im[dl, hh_, wl]
        if lo_d and hi_h and hi_w:
            weight += (dl + 1 - ad) * (aw - wl) * im[dl, hh_, wh_]
        if hi_d and lo_h and lo_w:
            weight += -1 * (ad - dl) * (wl + 1 - aw) * im[dh_, hl, wl]
        if hi_d and lo_h and hi_w:
            weight += -1 * (ad - dl) * (aw - wl) * im[dh_, hl, wh_]
        if hi_d and hi_h and lo_w:
            weight += (ad - dl) * (wl + 1 - aw) * im[dh_, hh_, wl]
        if hi_d and hi_h and hi_w:
            weight += (ad - dl) * (aw - wl) * im[dh_, hh_, wh_]
    else:
        if lo_d and lo_h and lo_w:
            weight += -1 * (dl + 1 - ad) * (hl + 1 - ah) * im[dl, hl, wl]
        if lo_d and lo_h and hi_w:
            weight += (dl + 1 - ad) * (hl + 1 - ah) * im[dl, hl, wh_]
        if lo_d and hi_h and lo_w:
            weight += -1 * (dl + 1 - ad) * (ah - hl) * im[dl, hh_, wl]
        if lo_d and hi_h and hi_w:
            weight += (dl + 1 - ad) * (ah - hl) * im[dl, hh_, wh_]
        if hi_d and lo_h and lo_w:
            weight += -1 * (ad - dl) * (hl + 1 - ah) * im[dh_, hl, wl]
        if hi_d and lo_h and hi_w:
            weight += (ad - dl) * (hl + 1 - ah) * im[dh_, hl, wh_]
        if hi_d and hi_h and lo_w:
            weight += -1 * (ad - dl) * (ah - hl) * im[dh_, hh_, wl]
        if hi_d and hi_h and hi_w:
            weight += (ad - dl) * (ah - hl) * im[dh_, hh_, wh_]
    return weight


def _sample_pos(offset, b, tap, kd, kh, kw, dc, hc, wc, stride, pad, dil):
    """(d_im, h_im, w_im) of tap (i, j, k) at output voxel (dc, hc, wc): cuh:224-247."""
    i, j, k = tap // (kh * kw), (tap // kw) % kh, tap % kw
    od, oh, ow = offset[b, 3 * tap, dc, hc, wc], offset[b, 3 * tap + 1, dc, hc, wc], offset[b, 3 * tap + 2, dc, hc, wc]
    return (dc * stride[0] - pad[0] + i * dil[0] + od, hc * stride[1] - pad[1] + j * dil[1] + oh, wc * stride[2] - pad[2] + k * dil[2] + ow)


def im2col(x, offset, ksize, stride, pad, dil):                          # cuh:192-265
    B, C, D, H, W, Do, Ho, Wo = _dims(x, ksize, stride, pad, dil)
    kd, kh, kw = ksize
    T = kd * kh * kw
    cols = np.zeros((C * T, B, Do, Ho, Wo))
    for c in range(C):
        for b in range(B):
            for dc in range(Do):
                for hc in range(Ho):
                    for wc in range(Wo):
                        for tap in range(T):
                            d_im, h_im, w_im = _sample_pos(offset, b, tap, kd, kh, kw, dc, hc, wc, stride, pad, dil)
                            val = 0.0
                            if d_im > -1 and h_im > -1 and w_im > -1 and d_im < D and h_im < H and w_im < W:     # cuh:248
                                val = _bilinear(x[b, c], D, H, W, d_im, h_im, w_im)
                            cols[c * T + tap, b, dc, hc, wc] = val
    return cols


def col2im(cols, offset, xshape, ksize, stride, pad, dil):              # cuh:267-334
    B, C, D, H, W = xshape
    kd, kh, kw = ksize
    T = kd * kh * kw
    _, _, Do, Ho, Wo = cols.shape
    grad_im = np.zeros(xshape)
    for c in range(C):
        for tap in range(T):
            for b in range(B):
                for dc in range(Do):
                    for hc in range(Ho):
                        for wc in range(Wo):
                            pd_, ph_, pw_ = _sample_pos(offset, b, tap, kd, kh, kw, dc, hc, wc, stride, pad, dil)
                            top = cols[c * T + tap, b, dc, hc, wc]
                            cd, ch, cw = int(pd_), int(ph_), int(pw_)          # C cast: truncation toward zero (cuh:303-305)
                            for dz in range(-2, 3):
                                for dy in range(-2, 3):
                                    for dx in range(-2, 3):
                                        z, y, xx = cd + dz, ch + dy, cw + dx
                                        if (0 <= z < D and 0 <= y < H and 0 <= xx < W and abs(pd_ - z) < 1 and abs(ph_ - y) < 1
                                                and abs(pw_ - xx) < 1):
                                            grad_im[b, c, z, y, xx] += _gradient_weight(pd_, ph_, pw_, z, y, xx, D, H, W) * top
    return grad_im


def col2im_coord(cols, x, offset, ksize, stride, pad, dil):              # cuh:336-405
    B, C, D, H, W, Do, Ho, Wo = _dims(x, ksize, stride, pad, dil)
    kd, kh, kw = ksize
    T = kd * kh * kw
    grad_offset = np.zeros((B, 3 * T, Do, Ho, Wo))
    for b in range(B):
        for oc in range(3 * T):
            tap, bp_dir = oc // 3, oc % 3
            for dc in range(Do):
                for hc in range(Ho):
                    for wc in range(Wo):
                        val = 0.0
                        for c in range(C):                                  # col_c = tap, tap + T, ... (cnt = c)
                            inv_d, inv_h, inv_w = _sample_pos(offset, b, tap, kd, kh, kw, dc, hc, wc, stride, pad, dil)
                            if inv_d <= -1 or inv_h <= -1 or inv_w <= -1 or inv_d >= D or inv_h >= H or inv_w >= W:
                                inv_d = inv_h = inv_w = -2
                            wgt = _coordinate_weight(inv_d, inv_h, inv_w, D, H, W, x[b, c], bp_dir)
                            val += wgt * cols[c * T + tap, b, dc, hc, wc]
                        grad_offset[b, oc, dc, hc, wc] = val
    return grad_offset


def forward(x, offset, weight, bias, stride=(1, 1, 1), pad=(1, 1, 1), dil=(1, 1, 1)):
    """out[b,k,p] = bias[k] + sum_{c,t} W[k,c,t] * col[c*T+t][b][p]   (deform_conv_cuda.cu:93-123)."""
    K, C = weight.shape[:2]
    ks = weight.shape[2:]
    cols = im2col(x, offset, ks, stride, pad, dil)
    out = np.einsum('kr,rbdhw->bkdhw', weight.reshape(K, -1), cols)
    return out + bias.reshape(1, K, 1, 1, 1)


def backward(x, offset, weight, bias, go, stride=(1, 1, 1), pad=(1, 1, 1), dil=(1, 1, 1)):
    """-> grad_input, grad_offset, grad_weight, grad_bias   (deform_conv_cuda.cu:226-277)."""
    K, C = weight.shape[:2]
    ks = weight.shape[2:]
    gcol = np.einsum('kr,bkdhw->rbdhw', weight.reshape(K, -1), go)               # cu:226-231
    grad_offset = col2im_coord(gcol, x, offset, ks, stride, pad, dil)
    grad_input = col2im(gcol, offset, x.shape, ks, stride, pad, dil)
    cols = im2col(x, offset, ks, stride, pad, dil)
    grad_weight = np.einsum('bkdhw,rbdhw->kr', go, cols).reshape(weight.shape)    # cu:254-271
    grad_bias = go.sum((0, 2, 3, 4))                                               # cu:277
    return grad_input, grad_offset, grad_weight, grad_bias
