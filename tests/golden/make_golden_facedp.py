#!/usr/bin/env python3
"""Golden vectors of the FaceDP data path made by RUNNING THE REFERENCE LOADER (build container only):

    python tests/golden/make_golden_facedp.py        ->  tests/golden/facedp_samples.json (+ facedp_sample0.npz)

The tiny on-disk dataset comes from tests/facedp_fixture.py (seeded); the reference's dataloader/FaceDP/loader.py is imported
from /root/reference and iterated with seeded RNGs.  Only data is saved: a shape/dtype/sha256 record per entry of every sample
dict, and the full arrays of one sample for debugging.  Nothing of the reference is copied; the GPU box never runs this.

Shims for third-party packages that this image lacks (published semantics, restated; parity is unpinned at exactly these points):
  cv2.cvtColor(COLOR_BGR2GRAY)                -- only feeds a mask the reference discards (path_reader.py:285 `normal, _`)
  torchvision.transforms.functional 0.6.0     -- to_tensor (u8 HWC -> f32 CHW / 255; float arrays: HWC -> CHW unchanged),
      (requirements.txt:4)                       normalize ((t - mean) / std in the tensor's dtype), adjust_brightness /
                                                 adjust_contrast (PIL ImageEnhance), adjust_gamma (255 * (v/255)^gamma point table)
"""
import importlib.util
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parents[2]
REF = '/root/reference'
OUT = REPO / 'tests' / 'golden'


def load_fixture_module():
    spec = importlib.util.spec_from_file_location('facedp_fixture', str(REPO / 'tests' / 'facedp_fixture.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_shims():
    cv2 = types.ModuleType('cv2')
    cv2.COLOR_BGR2GRAY = 6
    cv2.cvtColor = lambda img, code: img[..., 0] * 0.114 + img[..., 1] * 0.587 + img[..., 2] * 0.299
    sys.modules['cv2'] = cv2

    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    F = types.ModuleType('torchvision.transforms.functional')
    from PIL import Image, ImageEnhance

    def to_tensor(pic):
        if isinstance(pic, np.ndarray):
            if pic.ndim == 2:
                pic = pic[:, :, None]
            img = torch.from_numpy(pic.transpose((2, 0, 1)))
            if isinstance(img, torch.ByteTensor):
                return img.float().div(255)
            return img
        arr = np.asarray(pic)                                        # PIL image, 8-bit modes only here
        if arr.ndim == 2:
            arr = arr[:, :, None]
        img = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()
        return img.float().div(255) if isinstance(img, torch.ByteTensor) else img

    def normalize(tensor, mean, std, inplace=False):
        if not inplace:
            tensor = tensor.clone()
        mean = torch.as_tensor(mean, dtype=tensor.dtype)
        std = torch.as_tensor(std, dtype=tensor.dtype)
        tensor.sub_(mean[:, None, None]).div_(std[:, None, None])
        return tensor

    def adjust_brightness(img, factor):
        return ImageEnhance.Brightness(img).enhance(factor)

    def adjust_contrast(img, factor):
        return ImageEnhance.Contrast(img).enhance(factor)

    def adjust_gamma(img, gamma, gain=1):
        mode = img.mode
        img = img.convert('RGB')
        table = [int(255 * gain * pow(v / 255., gamma)) for v in range(256)] * 3
        return img.point(table).convert(mode)

    F.to_tensor, F.normalize = to_tensor, normalize
    F.adjust_brightness, F.adjust_contrast, F.adjust_gamma = adjust_brightness, adjust_contrast, adjust_gamma
    tvt.functional = F
    tv.transforms = tvt
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt, 'torchvision.transforms.functional': F})


def main():
    fx = load_fixture_module()
    install_shims()
    sys.path.insert(0, REF)
    record = {}
    full = {}
    home = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        datasets = {}
        for case, (body, training, ds_kwargs, seeds, count) in fx.CASES.items():
            key = json.dumps({k: str(v) for k, v in ds_kwargs.items()}, sort_keys=True)
            if key not in datasets:
                datasets[key] = fx.build_dataset(os.path.join(tmp, 'data%d' % len(datasets)), seed=0, **ds_kwargs)
            option, training = fx.make_option(case, datasets[key])
            work = os.path.join(tmp, 'cwd_' + case)                  # the reference writes its index cache into the cwd
            os.makedirs(work)
            os.chdir(work)
            try:
                from dataloader.FaceDP.loader import FaceDPLoader
                from dataloader.FaceDP.path_reader import RCV_DPreader
                # numpy >= 1.24 refuses the ragged np.save of loader.py:108, so the reference's own index builder is run here and
                # its result stored the way old numpy stored it (object array); the loader then takes its load branch (:110)
                entries, n = RCV_DPreader(option, option.dataset.path, training).read_rcv_path()
                holder = np.empty(2, dtype=object)
                holder[0], holder[1] = entries, n
                np.save('FaceDP_%s_%s.npy' % ('train' if training else 'test', 'multi' if option.use_multi else 'single'), holder)
                loader = FaceDPLoader(option, training)
                # the reference's index follows the file system's glob order; the port sorts -- compare on the sorted index
                loader.pathdata = sorted(loader.pathdata, key=lambda e: e['tar_view'])
                fx.seed_all(seeds)
                samples = []
                for i in range(count):
                    sample = loader[i]
                    samples.append({k: fx.digest(v) for k, v in sample.items()})
                    if case == 'train_soft_light' and i == 0:
                        full = {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sample.items()
                                if not isinstance(v, str)}
                record[case] = {'length': len(loader), 'samples': samples,
                                'index': [os.path.relpath(e['tar_view'], datasets[key]) for e in loader.pathdata]}
            finally:
                os.chdir(home)
    with open(OUT / 'facedp_samples.json', 'w') as fh:
        json.dump(record, fh, indent=1, sort_keys=True)
    np.savez_compressed(OUT / 'facedp_sample0.npz', **full)
    print('wrote', OUT / 'facedp_samples.json', {k: (v['length'], len(v['samples'])) for k, v in record.items()})


if __name__ == '__main__':
    main()
