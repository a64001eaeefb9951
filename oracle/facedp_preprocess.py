"""CPU oracle (TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this) of the FaceDP
per-sample arithmetic that csrc/dp_preprocess.hip runs on the device: numpy restatement of

  dataloader/FaceDP/path_reader.py:150-168   read_depth      mask = file mask > 0 or depth > 0, idepth = max(depth[mask]) / depth
  dataloader/FaceDP/path_reader.py:196-232   read_disparity  disp = fp64(a / depth + b) on the mask, 50 * max elsewhere / NaN / Inf
  dataloader/FaceDP/path_reader.py:299-301   casts           mask, disp, depth -> fp32 (idepth keeps the depth's dtype)
  dataloader/preprocess/augmentation.py:165-178  crop         window [y0:y0+ch, x0:x0+cw] of every input / target
  dataloader/preprocess/augmentation.py:62-84    ToTensor     u8 HWC -> fp32 CHW / 255, float HWC -> CHW unchanged, squeeze()
  dataloader/preprocess/augmentation.py:236-262  Lighting     + shift[c]
  dataloader/preprocess/augmentation.py:265-297  Normalizer   (t - mean[c]) / std[c] in fp32

Pinning: tests/test_facedp_host.py checks this file (fed by the product's host-side index / reader / draws) against
tests/golden/facedp_samples.json, which tests/golden/make_golden_facedp.py produced by running the reference's own loader on the
seeded tiny dataset of tests/facedp_fixture.py.  The reference's torchvision / cv2 calls went through restated shims there
(those packages are not in this image), so ToTensor / Normalizer / PIL photometric tables are pinned to torchvision 0.6.0's
published semantics, not to its binary.
"""
import numpy as np

MEAN = np.float32([0.485, 0.456, 0.406])
STD = np.float32([0.229, 0.224, 0.225])


def depth_targets(depth, file_mask, a, b):
    """Full-frame depth / mask / disp / idepth exactly as load_data_depth hands them to the transform."""
    depth = np.array(depth, copy=True)
    mask = (file_mask > 0) if file_mask is not None else (depth > 0)
    max_depth = np.max(depth[mask])
    idepth = np.zeros_like(depth)
    np.divide(max_depth, depth, out=idepth, where=mask)
    idepth[~mask] = 0.0
    depth[~mask] = 0.0
    disp = np.zeros(depth.shape, dtype=np.float64)
    np.divide(a, depth, out=disp, where=mask, dtype='float64')
    np.add(disp, b, out=disp, where=mask, dtype='float64')
    fill = np.max(disp[mask]) * 50.0
    disp[~mask] = fill
    disp[np.isnan(disp)] = fill
    disp[np.isinf(disp)] = fill
    return {'depth': depth.astype(np.float32), 'mask': mask.astype(np.float32), 'disp': disp.astype(np.float32), 'idepth': idepth}


def window(arr, x0, y0, ch, cw):
    return arr[y0:y0 + ch, x0:x0 + cw]


def image_tensor(img_u8, shift=None, lut=None, normalise=True):
    """u8 [h, w, 3] or [h, w] -> fp32 CHW following ToTensor -> Lighting -> Normalizer, every step rounded to fp32."""
    img = np.asarray(img_u8)
    if lut is not None:
        img = np.stack([lut[c][img[..., c]] for c in range(img.shape[2])], axis=2) if img.ndim == 3 else lut[0][img]
    grey = img.ndim == 2
    t = (img[None] if grey else img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    if not normalise:
        return t[0] if grey else t                                  # raw_transform: ToTensor + squeeze only
    if shift is not None and not grey:
        t = t + np.asarray(shift, dtype=np.float32).reshape(3, 1, 1)
    mean, std = (np.float32([0.5]), np.float32([0.5])) if grey else (MEAN, STD)
    return (t - mean.reshape(-1, 1, 1)) / std.reshape(-1, 1, 1)


def hwc_tensor(arr):
    arr = np.asarray(arr)
    return arr.transpose(2, 0, 1) if arr.ndim == 3 else arr


def sample_view(arrays, flags, win, shifts=None, luts=None, normalise=True, a=None, b=None):
    """arrays: dict with left / right / center (u8), depth, file_mask, normal, albedo as read from disk (None when unused);
    win = (x0, y0, ch, cw).  -> dict of numpy arrays named like the reference's sample dict."""
    x0, y0, ch, cw = win
    out = {}
    for name in ('left', 'right', 'center'):
        if arrays.get(name) is not None:
            out[name] = np.ascontiguousarray(image_tensor(window(arrays[name], x0, y0, ch, cw),
                                                          shift=(shifts or {}).get(name), lut=(luts or {}).get(name),
                                                          normalise=normalise))
    full = depth_targets(arrays['depth'], arrays.get('file_mask'), a, b)
    for name, flag in (('depth', 'depth'), ('mask', 'mask'), ('disp', 'disparity'), ('idepth', 'idepth')):
        if flags[flag]:
            out[name] = np.ascontiguousarray(window(full[name], x0, y0, ch, cw))
    for name in ('normal', 'albedo'):
        if arrays.get(name) is not None:
            out[name] = np.ascontiguousarray(hwc_tensor(window(arrays[name], x0, y0, ch, cw)))
    return out
