"""FaceDP on-disk data path: JSON index, file readers, augmentation draws on the host; crop, photometric
LUT, ToTensor, lighting noise, normalisation and the depth -> defocus-disparity conversion on the device (csrc/dp_preprocess.hip).

Reference behaviour mirrored here (nothing is imported from it):
  dataloader/FaceDP/path_reader.py:34-127   train.txt / test.txt -> JSON files filtered on INFO.{valid, lightidx, gender, camidx,
                                            expression, position, direction}; multi-view reference lists
  dataloader/FaceDP/path_reader.py:136-294  readers: images (the JSON's `right` file is the sample's `left`, :281), depth / mask /
                                            idepth, normal, albedo, disparity = a / depth + b from the per-camera table (:29-32),
                                            K / P assembly (:276-278, src/utils/geometry.py:144-167)
  dataloader/FaceDP/loader.py:131-197       sample dict: left/right/center, depth/mask/disp/idepth/normal/albedo, K/P/abvalue/
                                            metadata/coords, raw_* (use_raw), *s (use_multi), groupname / pathname
  dataloader/preprocess/preprocess.py:46-88 crop -> photometric -> ToTensor -> Lighting -> Normalizer, in this order of RNG draws:
                                            random.randint (row, then column), np.random.uniform per enabled photometric flag
                                            (brightness, gamma, contrast, light), one torch normal_(3) per image for the lighting

Design: the reference runs all of this per sample in DataLoader worker processes and ships finished fp32 tensors through shared
memory and a pinned copy (84 MB per 1024x1536 sample with use_raw).  Here decode threads produce the *raw* arrays (u8 images, the depth
/ normal .npy as stored: 38 MB), the draws are made in sample order on the consumer thread (reproducible for any number of decode
threads), the raw arrays go to HBM once over a side stream and the fused kernels write straight into the batch tensors.

There is no CPU compute path: `DevicePreprocessor` needs libdpf_hip.so and a GPU (the numpy restatement lives in
oracle/facedp_preprocess.py and is test infrastructure).
"""
import ast
import ctypes
import json
import math
import os
import random
import re
import warnings
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

# affine depth -> disparity parameters [a, b] per camera index (path_reader.py:29-32; calibration data of the FaceDP rig)
ABVALUE_BY_CAMERA = {
    1: (-26996.48848727, 32.984822), 2: (-25727.48737484, 31.80317696), 3: (-24940.24188275, 30.52371982),
    4: (-25821.86619949, 32.03359466), 5: (-26735.69581971, 33.24327157), 6: (-22694.45143825, 27.76217617),
    7: (-23598.82548605, 29.1246567), 8: (-26482.94764346, 32.91372342),
}
# (focal length mm, focused distance mm, f-number, pixel size um), path_reader.py:264
METADATA = (135.0, 970.0, 5.657, 5.36)
IMAGENET_MEAN = (0.485, 0.456, 0.406)          # augmentation.py:279-280
IMAGENET_STD = (0.229, 0.224, 0.225)
# AlexNet-style PCA lighting basis (augmentation.py:241-244)
LIGHT_EIGVAL = (0.2175, 0.0188, 0.0045)
LIGHT_EIGVEC = ((-0.5675, 0.7192, 0.4009), (-0.5808, -0.0045, -0.8140), (-0.5836, -0.6948, 0.4203))

_INFO_FILTERS = (('lightidx', 'light'), ('gender', 'gender'), ('camidx', 'viewpoint'), ('expression', 'expression'),
                 ('position', 'distance'), ('direction', 'direction'))


def _get(opt, name, default=None):
    return getattr(opt, name, default) if opt is not None else default


# ------------------------------------------------------------------------------------------------------------------ index
def parse_array_literal(text):
    """'array([...])' (numpy repr stored in the JSON, path_reader.py:240-254) -> nested list, without eval."""
    if text is None:
        return None
    m = re.search(r'\[.*\]', text, flags=re.S)
    if not m:
        raise ValueError('not an array literal: %r' % (text,))
    return ast.literal_eval(m.group(0))


def read_split(parentdir, training):
    """train.txt / test.txt -> list of subject directories (path_reader.py:34-54)."""
    listing = Path(parentdir) / ('train.txt' if training else 'test.txt')
    if not listing.is_file():
        raise FileNotFoundError('%s does not exist.' % listing)
    with open(str(listing), 'r') as fh:
        return [Path(parentdir) / ln.replace('\n', '').replace('\r', '') for ln in fh if ln.strip()]


def build_index(option, parentdir, training):
    """-> list of {'tar_view', 'ref_view', 'parentdir'} entries (path_reader.py:56-127).  JSON files are visited in sorted order
    (the reference takes the file system's order, which is not reproducible across machines)."""
    data_opt = option.dataset
    use_multi = bool(_get(option, 'use_multi', False))
    entries = []
    for subject in read_split(parentdir, training):
        jdir = subject / 'JSON'
        for jpath in sorted(jdir.glob('*.json')):
            with open(str(jpath)) as fh:
                info = json.load(fh)['INFO']
            if not bool(info['valid']):
                continue
            if any(info[key] not in getattr(data_opt, optname) for key, optname in _INFO_FILTERS):
                continue
            entry = {'ref_view': None, 'tar_view': str(jpath), 'parentdir': str(subject)}
            if use_multi:
                view = int(str(jpath).split('_')[-2])
                light = int(info['lightidx'])
                refs = []
                for cam in data_opt.select_view:
                    ref = jdir / ('INFO_%d_%d_%d.json' % (cam, view, light))
                    with open(str(ref)) as fh:
                        if bool(json.load(fh)['INFO']['valid']):
                            refs.append(str(ref))
                if not refs:
                    continue
                refs += [refs[-1]] * (len(data_opt.select_view) - len(refs))      # pad with the last valid view (:107-109)
                entry['ref_view'] = refs
            entries.append(entry)
    return entries


def cache_name(option, training):
    """loader.py:93-102"""
    return '%s_%s_%s.npy' % (option.dataset_name, 'train' if training else 'test', 'multi' if _get(option, 'use_multi', False) else 'single')


def load_or_build_index(option, parentdir, training, cache_dir='.'):
    """The reference caches the index as a pickled object array in the working directory (loader.py:105-110); the same file is
    read and written here so an index built by either side serves both."""
    path = os.path.join(cache_dir, cache_name(option, training))
    if os.path.isfile(path):
        entries, _ = np.load(path, allow_pickle=True)
        return list(entries)
    entries = build_index(option, parentdir, training)
    holder = np.empty(2, dtype=object)
    holder[0], holder[1] = entries, len(entries)
    np.save(path, holder)
    return entries


# ------------------------------------------------------------------------------------------------------------------ reader
class RawSample(object):
    """Decoded files of one view, full resolution, host memory.  images: u8 [H, W, 3] (or [H, W]); depth as stored (fp32 / fp64)."""
    __slots__ = ('left', 'right', 'center', 'depth', 'file_mask', 'normal', 'albedo', 'K', 'P', 'abvalue', 'metadata', 'a', 'b')

    def image_shape(self):
        for arr in (self.left, self.right, self.center, self.depth):
            if arr is not None:
                return int(arr.shape[0]), int(arr.shape[1])
        raise ValueError('sample without any image')


def _view_flags(option, multi):
    src = option.multi_view if multi else option
    return {k: bool(_get(src, 'use_' + k, False)) for k in
            ('dual_pixel', 'center_img', 'mask', 'disparity', 'depth', 'idepth', 'normal', 'albedo')}


def read_raw(json_data, parentdir, option, multi=False):
    """Files of one JSON entry (path_reader.py:266-351 load_data_depth without the arithmetic, which runs on the device)."""
    from PIL import Image
    root = Path(parentdir)
    flags = _view_flags(option, multi)
    s = RawSample()
    intrinsic = parse_array_literal(json_data['PARAMS']['intrinsic'])
    extrinsic = np.asarray(parse_array_literal(json_data['PARAMS']['pose']), dtype=np.float64).reshape(-1)
    K = np.zeros((3, 3))
    K[0, 0], K[0, 1], K[0, 2], K[1, 1], K[1, 2], K[2, 2] = intrinsic[0], intrinsic[2], intrinsic[3], intrinsic[1], intrinsic[4], 1
    P = np.concatenate([extrinsic, np.zeros(3), np.ones(1)]).reshape(4, 4)
    a, b = ABVALUE_BY_CAMERA[json_data['INFO']['camidx']]
    s.K, s.P = np.float32(K), np.float32(P)
    s.a, s.b = float(a), float(b)
    s.abvalue = np.float32([b, a])                                   # returned swapped, path_reader.py:207
    s.metadata = np.float32(METADATA)

    def image(key):
        return np.asarray(Image.open(str(root / json_data['PATH'][key])))

    # the reference unpacks (left, right, lr) into (right, left, lr): the file stored as `right` is the network's left view (:281)
    s.left = image('right') if flags['dual_pixel'] else None
    s.right = image('left') if flags['dual_pixel'] else None
    s.center = image('lrsum') if flags['center_img'] else None
    s.depth = np.load(str(root / json_data['PATH']['depth']))
    if s.depth.dtype not in (np.float32, np.float64):
        s.depth = s.depth.astype(np.float32)
    s.file_mask = None
    if 'mask' in json_data['PATH']:
        s.file_mask = np.ascontiguousarray(np.load(str(root / json_data['PATH']['mask'])) > 0).view(np.uint8)
    s.normal = np.ascontiguousarray(np.load(str(root / json_data['PATH']['normal'])), dtype=np.float32) if flags['normal'] else None
    s.albedo = np.ascontiguousarray(np.load(str(root / json_data['PATH']['albedo'])), dtype=np.float32) if flags['albedo'] else None
    return s, flags


# ------------------------------------------------------------------------------------------------------------------ draws
def crop_size(shape, crop_opt):
    """preprocess.py:26-41,60-64: soft crop = the largest multiple of crop_factor not above ratio * size; hard crop = fixed."""
    if crop_opt.type == 'soft_crop':
        ratio, factor = crop_opt.soft_crop.crop_ratio, crop_opt.soft_crop.crop_factor
        n = np.ceil(np.array(shape[:2]) * ratio // factor).astype('int')
        return int(factor * n[0]), int(factor * n[1])
    return int(crop_opt.hard_crop.crop_height), int(crop_opt.hard_crop.crop_width)


def draw_crop(shape, size, crop_opt, mask=None):
    """-> (x0, y0): augmentation.py:120-199.  Consumes python `random` exactly like the reference (row first, then column)."""
    h, w = shape[:2]
    th, tw = size
    method = crop_opt.method
    if method == 'mask_random_crop' and mask is None:
        method = 'random_crop'
    if method == 'center_crop':
        return int(round((w - tw) / 2.)), int(round((h - th) / 2.))
    if method == 'random_crop':
        y0 = random.randint(0, h - th)
        x0 = random.randint(0, w - tw)
        return x0, y0
    if method == 'mask_random_crop':
        ys, xs = np.nonzero(mask > 0)
        roiy, roix = int(ys.min()), int(xs.min())
        trial = 0
        while True:
            y0 = random.randint(roiy, h - th)
            x0 = random.randint(roix, w - tw)
            if np.sum(mask[y0:y0 + th, x0:x0 + tw]) / (th * tw) >= crop_opt.min_inlier:
                return x0, y0
            trial += 1
            if trial >= crop_opt.max_trial:
                y0 = random.randint(0, h - th)
                x0 = random.randint(0, w - tw)
                return x0, y0
    raise NotImplementedError('invalid cropping method')


def lighting_shift(alphastd):
    """One Lighting draw (augmentation.py:250-261): alpha ~ N(0, alphastd)^3 from the torch RNG, shift = sum_k eigvec[:, k] * alpha_k
    * eigval_k, evaluated with the same fp32 torch ops so the value is bit-identical."""
    alpha = torch.empty(3).normal_(0, alphastd)
    eigvec = torch.tensor(LIGHT_EIGVEC)
    eigval = torch.tensor(LIGHT_EIGVAL)
    return eigvec.mul(alpha.view(1, 3).expand(3, 3)).mul(eigval.view(1, 3).expand(3, 3)).sum(1)


def photometric_lut(image_u8, brightness, gamma, contrast):
    """u8 [C, 256] table equal to torchvision's PIL adjust_brightness -> adjust_gamma -> adjust_contrast chain
    (augmentation.py:222-232; a factor of 0 skips the step).  Brightness and contrast are PIL ImageEnhance blends against black / the
    rounded mean of the grey image, gamma is a 256-entry point table, so the chain is a per-value map once the grey mean is known."""
    from PIL import Image, ImageEnhance
    C = 1 if image_u8.ndim == 2 else image_u8.shape[2]
    ramp = np.arange(256, dtype=np.uint8)
    table = np.repeat(ramp[None, :, None], C, axis=2) if C > 1 else ramp[None, :]          # [1, 256, C] ramp "image"
    pil = Image.fromarray(np.ascontiguousarray(table))
    ref = Image.fromarray(image_u8)
    if brightness != 0:
        pil = ImageEnhance.Brightness(pil).enhance(brightness)
        ref = ImageEnhance.Brightness(ref).enhance(brightness) if contrast != 0 else ref
    if gamma != 0:
        gmap = [int(255 * pow(v / 255., gamma)) for v in range(256)] * C          # torchvision 0.6.0 (requirements.txt:4) table, truncated
        pil = pil.point(gmap)
        ref = ref.point(gmap) if contrast != 0 else ref
    if contrast != 0:
        from PIL import ImageStat
        mean = int(ImageStat.Stat(ref.convert('L')).mean[0] + 0.5)      # the degenerate image depends on the actual picture
        degenerate = Image.new('L', pil.size, mean).convert(pil.mode)
        pil = Image.blend(degenerate, pil, contrast)
    out = np.asarray(pil)
    return np.ascontiguousarray(out.reshape(256, C).T if C > 1 else out.reshape(1, 256))


class SamplePlan(object):
    """The random decisions of one sample: crop window and, per image, lighting shift and photometric table."""
    __slots__ = ('x0', 'y0', 'ch', 'cw', 'shift', 'lut', 'normalise')


def draw_plan(option, raw, flags):
    """preprocess.py:46-88, in the reference's draw order."""
    aug = list(_get(option, 'augmentation', []) or [])
    h, w = raw.image_shape()
    plan = SamplePlan()
    plan.x0, plan.y0, plan.ch, plan.cw = 0, 0, h, w
    plan.shift = {'left': None, 'right': None, 'center': None}
    plan.lut = {'left': None, 'right': None, 'center': None}
    plan.normalise = True
    if 'crop_aug' in aug:
        copt = option.crop_aug
        plan.ch, plan.cw = crop_size((h, w), copt)
        mask = None
        if copt.method == 'mask_random_crop' and flags['mask']:
            mask = (raw.file_mask > 0) if raw.file_mask is not None else (raw.depth > 0)
        plan.x0, plan.y0 = draw_crop((h, w), (plan.ch, plan.cw), copt, mask)
    if 'photo_aug' in aug:
        popt = option.photo_aug
        brightness = np.random.uniform(0.7, 1.2, 1)[0] if popt.brightness else 0
        gamma = np.random.uniform(0.7, 1.2, 1)[0] if popt.gamma else 0
        contrast = np.random.uniform(0.7, 1.2, 1)[0] if popt.contrast else 0
        light = np.random.uniform(0.5, 5.0, 1)[0] if popt.light else 0
        for name in ('left', 'right', 'center'):
            img = getattr(raw, name)
            if img is None:
                continue
            if brightness != 0 or gamma != 0 or contrast != 0:
                window = img[plan.y0:plan.y0 + plan.ch, plan.x0:plan.x0 + plan.cw]
                plan.lut[name] = photometric_lut(np.ascontiguousarray(window), brightness, gamma, contrast)
        for name in ('left', 'right', 'center'):                     # Lighting runs after ToTensor over all inputs (:76-80)
            img = getattr(raw, name)
            if img is not None and light != 0 and img.ndim == 3:
                plan.shift[name] = lighting_shift(light)
    return plan


# ------------------------------------------------------------------------------------------------------------------ device
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _floats(values):
    return (ctypes.c_float * len(values))(*[float(v) for v in values])


class DevicePreprocessor(object):
    """Uploads one RawSample and fills views of it (a crop with augmentation, the raw full frame) into caller tensors.  All work is
    enqueued on `stream` (default: the current stream)."""

    def __init__(self, device):
        from ._lib import lib
        self.lib = lib()                                           # raises when libdpf_hip.so is missing: no CPU fallback
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('FaceDP preprocessing runs on the GPU only (got device %s)' % (device,))
        self.stats_doubles = int(self.lib.call('dpf_dp_stats_doubles'))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def upload(self, raw, pinned=None):
        """-> dict of device tensors (+ the stats scratch with the full-frame depth / disparity maxima already enqueued)."""
        dev = {}
        for name in ('left', 'right', 'center', 'depth', 'file_mask', 'normal', 'albedo'):
            arr = getattr(raw, name)
            if arr is None:
                dev[name] = None
                continue
            with warnings.catch_warnings():                           # PIL hands out read-only arrays; they are only read here
                warnings.simplefilter('ignore', UserWarning)
                host = torch.from_numpy(np.ascontiguousarray(arr))
            if pinned is not None:
                host = pinned.stage(name, host)
            dev[name] = host.to(self.device, non_blocking=True)
        stats = torch.empty(self.stats_doubles, dtype=torch.float64, device=self.device)
        d = dev['depth']
        self.lib.call('dpf_dp_depth_stats', _ptr(d), int(d.dtype == torch.float64), _ptr(dev['file_mask']), d.numel(), raw.a, raw.b,
                      _ptr(stats), self._stream())
        dev['stats'] = stats
        return dev

    def image(self, src, out, window, shift=None, lut=None, normalise=True):
        x0, y0, ch, cw = window
        H, W = src.shape[0], src.shape[1]
        C = 1 if src.dim() == 2 else src.shape[2]
        if normalise:
            mean, std = (IMAGENET_MEAN, IMAGENET_STD) if C == 3 else ((0.5,), (0.5,))       # augmentation.py:296-302
        else:
            mean, std = (0.0,) * C, (1.0,) * C
        # torch builds the mean / std / shift tensors in fp32; round the same way before the kernel sees them
        f32 = lambda seq: [float(np.float32(v)) for v in seq]
        if C == 1:
            shift = None                                               # Lighting only touches 3-channel tensors (augmentation.py:253)
        lut_dev = None
        if lut is not None:
            lut_dev = torch.from_numpy(np.ascontiguousarray(lut)).to(self.device, non_blocking=True)
        self.lib.call('dpf_dp_image', _ptr(src), _ptr(lut_dev), _ptr(out), H, W, C, y0, x0, ch, cw,
                      _floats(f32(shift.tolist())) if shift is not None else None, _floats(f32(mean)), _floats(f32(std)), self._stream())
        return lut_dev                                             # caller keeps it alive until the stream has consumed it

    def targets(self, dev, raw, window, depth=None, mask=None, disp=None, idepth=None):
        x0, y0, ch, cw = window
        d = dev['depth']
        self.lib.call('dpf_dp_targets', _ptr(d), int(d.dtype == torch.float64), _ptr(dev['file_mask']), _ptr(dev['stats']), raw.a, raw.b,
                      d.shape[0], d.shape[1], y0, x0, ch, cw, _ptr(depth), _ptr(mask), _ptr(disp), _ptr(idepth), self._stream())

    def hwc_to_chw(self, src, out, window):
        x0, y0, ch, cw = window
        self.lib.call('dpf_dp_hwc_to_chw', _ptr(src), _ptr(out), src.shape[0], src.shape[1], src.shape[2], y0, x0, ch, cw, self._stream())


_IMAGES = ('left', 'right', 'center')
_TARGETS = ('depth', 'mask', 'disp', 'idepth', 'normal', 'albedo')


def _allocate_view(dev, flags, ch, cw, device, prefix='', batch=None):
    """Empty output tensors of one view ([C, ch, cw] / [ch, cw]), or of a batch when `batch` is given."""
    lead = () if batch is None else (batch,)
    out = {}
    for name in _IMAGES:
        src = dev[name]
        if src is not None:                                           # a grey image leaves the Normalizer as [1, h, w] (:293-296)
            grey = (ch, cw) if prefix == 'raw_' else (1, ch, cw)
            out[prefix + name] = torch.empty(lead + ((3, ch, cw) if src.dim() == 3 else grey), device=device)
    if flags['depth']:
        out[prefix + 'depth'] = torch.empty(lead + (ch, cw), device=device)
    if flags['mask']:
        out[prefix + 'mask'] = torch.empty(lead + (ch, cw), device=device)
    if flags['disparity']:
        out[prefix + 'disp'] = torch.empty(lead + (ch, cw), device=device)
    if flags['idepth']:
        out[prefix + 'idepth'] = torch.empty(lead + (ch, cw), device=device, dtype=dev['depth'].dtype)
    for name in ('normal', 'albedo'):
        if dev[name] is not None:
            src = dev[name]
            out[prefix + name] = torch.empty(lead + ((src.shape[2], ch, cw) if src.dim() == 3 else (ch, cw)), device=device)
    return out


def _fill_view(pre, dev, raw, flags, window, out, prefix, plan=None, keep=None):
    """Run the kernels of one view into `out[prefix + name]` (already sliced to this sample)."""
    for name in _IMAGES:
        if dev[name] is None:
            continue
        shift = plan.shift[name] if plan is not None else None
        lut = plan.lut[name] if plan is not None else None
        held = pre.image(dev[name], out[prefix + name], window, shift=shift, lut=lut, normalise=plan is not None)
        if held is not None and keep is not None:
            keep.append(held)
    g = lambda n: out.get(prefix + n)
    if any(g(n) is not None for n in ('depth', 'mask', 'disp', 'idepth')):
        pre.targets(dev, raw, window, depth=g('depth'), mask=g('mask'), disp=g('disp'), idepth=g('idepth'))
    for name in ('normal', 'albedo'):
        if dev[name] is not None:
            if dev[name].dim() == 3:
                pre.hwc_to_chw(dev[name], out[prefix + name], window)
            else:
                x0, y0, ch, cw = window
                out[prefix + name].copy_(dev[name][y0:y0 + ch, x0:x0 + cw])


class FaceDPLoader(torch.utils.data.Dataset):
    """Drop-in for the reference's `FaceDPLoader(option, training)` (dataloader/FaceDP/loader.py:79-200): `__getitem__` returns the
    same sample dict, with the tensors resident on `device`.  Iterate it through `FaceDPBatcher` (below) rather than a
    multi-process DataLoader: the samples are produced on the GPU."""

    def __init__(self, option, training, device=None, cache_dir='.'):
        self.opt = option
        self.training = training
        self.parentdir = option.dataset.path
        self.use_multi = bool(_get(option, 'use_multi', False))
        if not os.path.isdir(self.parentdir):
            raise FileNotFoundError('%s does not exist.' % self.parentdir)
        self.pathdata = load_or_build_index(option, self.parentdir, training, cache_dir)
        self.device = device
        self._pre = None

    def __len__(self):
        return len(self.pathdata)

    # -- host half: files of one index entry (thread-safe, no RNG)
    def read(self, index):
        entry = self.pathdata[index]
        with open(entry['tar_view']) as fh:
            raw, flags = read_raw(json.load(fh), entry['parentdir'], self.opt)
        refs = []
        if self.use_multi:
            if entry['ref_view'] is None:
                raise RuntimeError('multi-view dataloader error')
            for ref in entry['ref_view']:
                with open(ref) as fh:
                    refs.append(read_raw(json.load(fh), entry['parentdir'], self.opt, multi=True))
        return raw, flags, refs

    def names(self, index):
        tar = self.pathdata[index]['tar_view']
        out = {'pathname': os.path.split(tar)[-1].split('.')[0]}
        if not self.training:
            out['groupname'] = tar.split('/')[-3]
        return out

    def preprocessor(self):
        if self._pre is None:
            device = self.device if self.device is not None else torch.device('cuda', torch.cuda.current_device())
            self._pre = DevicePreprocessor(device)
        return self._pre

    # -- device half
    def produce(self, index, item, out=None, slot=None, keep=None):
        """Draw the plan and enqueue the kernels of one sample.  `out` / `slot`: batch tensors and the row to fill (allocated per
        sample when None).  Returns (sample dict of tensors / host params, stats tensor)."""
        raw, flags, refs = item
        pre = self.preprocessor()
        plan = draw_plan(self.opt, raw, flags)
        dev = pre.upload(raw)
        window = (plan.x0, plan.y0, plan.ch, plan.cw)
        sample = {}
        view = _allocate_view(dev, flags, plan.ch, plan.cw, pre.device) if out is None else {k: v[slot] for k, v in out.items() if not k.startswith('raw_')}
        _fill_view(pre, dev, raw, flags, window, view, '', plan=plan, keep=keep)
        sample.update(view)
        K = raw.K.copy()
        K[0, 2] -= plan.x0                                          # loader.py:158-159
        K[1, 2] -= plan.y0
        sample.update({'K': K, 'P': raw.P, 'abvalue': raw.abvalue, 'metadata': raw.metadata,
                       'coords': np.asarray([plan.x0, plan.y0])})     # add2output turns the [x0, y0] list into an int64 array
        if bool(_get(self.opt, 'use_raw', False)):
            h, w = raw.image_shape()
            rview = (_allocate_view(dev, flags, h, w, pre.device, 'raw_') if out is None
                     else {k: v[slot] for k, v in out.items() if k.startswith('raw_')})
            _fill_view(pre, dev, raw, flags, (0, 0, h, w), rview, 'raw_', plan=None)
            sample.update(rview)
        stats = [dev['stats']]
        if self.use_multi:
            per_view = []
            for rraw, rflags in refs:
                rdev = pre.upload(rraw)
                h, w = rraw.image_shape()
                rv = _allocate_view(rdev, rflags, h, w, pre.device)
                _fill_view(pre, rdev, rraw, rflags, (0, 0, h, w), rv, '', plan=None)
                stats.append(rdev['stats'])
                per_view.append((rv, rraw))
            for name in _IMAGES + _TARGETS:                          # loader.py:184-193: tensors are concatenated along dim 0
                if name in per_view[0][0]:
                    sample[name + 's'] = torch.cat([rv[name] for rv, _ in per_view], dim=0)
            for name, attr in (('Ks', 'K'), ('Ps', 'P'), ('abvalues', 'abvalue'), ('metadatas', 'metadata')):
                sample[name] = np.asarray([getattr(rraw, attr) for _, rraw in per_view])
        sample.update(self.names(index))
        if keep is not None:
            keep.append(dev)
        return sample, stats

    def __getitem__(self, index):
        sample, stats = self.produce(index, self.read(index))
        check_stats(stats, sample['pathname'])
        return sample


def check_stats(stats, what=''):
    """The reference raises on NaN / Inf in the produced maps (path_reader.py:230, preprocess.py:85-86); the kernels count them."""
    for st in stats:
        head = st[:4].cpu()
        if float(head[3]) == 0:
            raise ValueError('FaceDP sample %s has no valid depth pixel' % what)
        if float(head[2]) != 0 or not math.isfinite(float(head[0])) or not math.isfinite(float(head[1])):
            raise ValueError('Nan or inf value is detected in the depth / disparity map of %s' % what)


def collate_params(samples):
    """default_collate for the host-side entries of the sample dicts."""
    out = {}
    for key in ('K', 'P', 'abvalue', 'metadata', 'Ks', 'Ps', 'abvalues', 'metadatas'):
        if key in samples[0]:
            out[key] = torch.from_numpy(np.stack([np.asarray(s[key]) for s in samples]))
    out['coords'] = torch.from_numpy(np.stack([s['coords'] for s in samples]))
    for key in ('pathname', 'groupname'):
        if key in samples[0]:
            out[key] = [s[key] for s in samples]
    return out


class FaceDPBatcher(object):
    """Iterable of device-resident batches over a FaceDPLoader: `workers` threads read and decode files ahead of the consumer, the
    consumer thread draws the augmentation plan in sample order, uploads on a side stream and runs the preprocessing kernels directly
    into the batch tensors; the training stream only waits on the batch's event.  Stands where the reference has
    torch.utils.data.DataLoader(loader, batch_size, shuffle, num_workers, pin_memory) (src/model/model.py dataloader hooks)."""

    def __init__(self, dataset, batch_size, shuffle=False, workers=4, drop_last=False, sampler=None, seed=1, prefetch_batches=2):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), bool(shuffle), bool(drop_last)
        self.num_workers = max(1, int(workers))
        self.sampler, self.seed, self.epoch = sampler, int(seed), 0
        self.prefetch = max(1, int(prefetch_batches)) * self.batch_size
        self.pin_memory, self.collate_fn = False, None              # DataLoader attributes some callers read
        self._side = None

    def with_sampler(self, sampler):
        return FaceDPBatcher(self.dataset, self.batch_size, self.shuffle, self.num_workers, self.drop_last, sampler, self.seed,
                             self.prefetch // self.batch_size)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)
        if self.sampler is not None and hasattr(self.sampler, 'set_epoch'):
            self.sampler.set_epoch(epoch)

    def _order(self):
        if self.sampler is not None:
            return list(iter(self.sampler))
        n = len(self.dataset)
        if not self.shuffle:
            return list(range(n))
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)                       # own generator: independent of the global RNG state
        return torch.randperm(n, generator=g).tolist()

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        """Batches are produced by one background thread (file reads fan out to `workers` decode threads behind it), so uploads and
        kernels of batch i + 1 run while the caller trains on batch i.  The augmentation draws all happen in that one thread, in
        sample order: for fixed seeds of `random`, `numpy.random` and torch the stream of crops is reproducible."""
        import queue
        import threading
        pre = self.dataset.preprocessor()
        if self._side is None:
            self._side = torch.cuda.Stream(device=pre.device)
        order = self._order()
        ready = queue.Queue(maxsize=max(1, self.prefetch // self.batch_size))
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def producer():
            try:
                torch.cuda.set_device(pre.device)
                with ThreadPoolExecutor(self.num_workers) as pool:
                    pending, cursor = [], 0
                    while not stop.is_set():
                        while cursor < len(order) and len(pending) < self.prefetch + self.batch_size:
                            pending.append((order[cursor], pool.submit(self.dataset.read, order[cursor])))
                            cursor += 1
                        take = pending[:self.batch_size]
                        if not take or (len(take) < self.batch_size and self.drop_last):
                            break
                        del pending[:len(take)]
                        if not put(('batch', self._assemble(pre, [(idx, fut.result()) for idx, fut in take]))):
                            break
                    for _, fut in pending:
                        fut.cancel()
                put(('end', None))
            except BaseException as exc:                              # surfaces in the consumer
                put(('error', exc))

        thread = threading.Thread(target=producer, name='facedp-batcher', daemon=True)
        thread.start()
        try:
            while True:
                kind, payload = ready.get()
                if kind == 'end':
                    return
                if kind == 'error':
                    raise payload
                batch, done = payload
                current = torch.cuda.current_stream(pre.device)
                current.wait_event(done)
                for v in batch.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(current)
                yield batch
        finally:
            stop.set()
            thread.join()

    def _assemble(self, pre, items):
        ds = self.dataset
        B = len(items)
        keep, samples, all_stats = [], [], []
        with torch.cuda.stream(self._side):
            out = None
            for slot, (idx, item) in enumerate(items):
                if out is None:
                    out = self._allocate(pre, item, B)
                sample, stats = ds.produce(idx, item, out=out, slot=slot, keep=keep)
                samples.append(sample)
                all_stats += stats
            heads = torch.stack([st[:4] for st in all_stats]).cpu()     # also drains the side stream before `keep` is released
            done = torch.cuda.Event()
            done.record(self._side)
        for row, sample in zip(heads.view(len(samples), -1, 4), samples):
            for head in row:
                if float(head[3]) == 0 or float(head[2]) != 0 or not math.isfinite(float(head[0])) or not math.isfinite(float(head[1])):
                    raise ValueError('Nan or inf value is detected in the maps of %s' % sample['pathname'])
        batch = dict(out)
        for key in samples[0]:
            if key.endswith('s') and torch.is_tensor(samples[0][key]) and key not in batch:      # multi-view stacks
                batch[key] = torch.stack([s[key] for s in samples])
        batch.update(collate_params(samples))
        return batch, done

    def _allocate(self, pre, item, B):
        raw, flags, _ = item
        plan_h, plan_w = raw.image_shape()
        aug = list(_get(self.dataset.opt, 'augmentation', []) or [])
        ch, cw = crop_size((plan_h, plan_w), self.dataset.opt.crop_aug) if 'crop_aug' in aug else (plan_h, plan_w)
        probe = {n: (torch.empty((0,) * getattr(raw, n).ndim) if getattr(raw, n) is not None else None)
                 for n in ('left', 'right', 'center', 'normal', 'albedo')}
        probe['depth'] = torch.empty(0, dtype=torch.float64 if raw.depth.dtype == np.float64 else torch.float32)
        for n in ('normal', 'albedo'):
            if getattr(raw, n) is not None and getattr(raw, n).ndim == 3:
                probe[n] = torch.empty((0, 0, getattr(raw, n).shape[2]))
        out = _allocate_view(probe, flags, ch, cw, pre.device, '', batch=B)
        if bool(_get(self.dataset.opt, 'use_raw', False)):
            out.update(_allocate_view(probe, flags, plan_h, plan_w, pre.device, 'raw_', batch=B))
        return out
