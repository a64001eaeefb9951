"""A tiny FaceDP-format dataset written to disk from a seed (test data, shared by the golden generator and the tests), and the
option objects of the test cases.  Layout and JSON fields follow the format description in the reference's
dataloader/FaceDP/loader.py:15-76 (docstring) as consumed by path_reader.py."""
import json
import os

import numpy as np

H, W = 48, 64


class Opt(object):
    def __init__(self, d):
        for k, v in d.items():
            setattr(self, k, Opt(v) if isinstance(v, dict) else v)


def _array_repr(values):
    return repr(np.asarray(values, dtype=np.float64))


def build_dataset(root, seed=0, depth_dtype=np.float32, mask_file=False, grey=False):
    """Two training subjects and one test subject, cameras 1/2/6, two head views, lights 1/2; some entries invalid or filtered."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    root = str(root)
    os.makedirs(root, exist_ok=True)
    subjects = {'train': ['s01_w', 's02_m'], 'test': ['s03_w']}
    for split, names in subjects.items():
        with open(os.path.join(root, split + '.txt'), 'w') as fh:
            fh.write('\n'.join(names) + '\n')
    expressions = ['neutral', 'frowning', 'smiling']
    count = 0
    for names in subjects.values():
        for name in names:
            base = os.path.join(root, name)
            for sub in ('JSON', 'IMG/LEFT', 'IMG/RIGHT', 'IMG/LRSUM', 'DEPTH', 'NORMAL', 'ALBEDO', 'MASK'):
                os.makedirs(os.path.join(base, sub), exist_ok=True)
            for cam in (1, 2, 6):
                for view in (0, 1):
                    yy, xx = np.mgrid[0:H, 0:W]
                    cy, cx = H * rng.uniform(0.35, 0.65), W * rng.uniform(0.35, 0.65)
                    rad = np.hypot((yy - cy) / (0.42 * H), (xx - cx) / (0.42 * W))
                    face = rad < 1.0
                    depth = np.where(face, 950.0 + 120.0 * rad ** 2 + rng.uniform(-3, 3, (H, W)), 0.0).astype(depth_dtype)
                    normal = rng.normal(size=(H, W, 3))
                    normal /= np.linalg.norm(normal, axis=2, keepdims=True)
                    normal = np.where(face[..., None], normal, 0.0).astype(np.float32)
                    albedo = np.where(face[..., None], rng.uniform(0.1, 1.0, (H, W, 3)), 0.0).astype(np.float32)
                    np.save(os.path.join(base, 'DEPTH', 'DEPTH_%d_%d.npy' % (cam, view)), depth)
                    np.save(os.path.join(base, 'NORMAL', 'NORMAL_%d_%d.npy' % (cam, view)), normal)
                    np.save(os.path.join(base, 'ALBEDO', 'ALBEDO_%d_%d.npy' % (cam, view)), albedo)
                    if mask_file:
                        np.save(os.path.join(base, 'MASK', 'MASK_%d_%d.npy' % (cam, view)), (face & (rad < 0.9)).astype(np.uint8))
                    for light in (1, 2):
                        stem = 'IMG_%d_%d_%d.png' % (cam, view, light)
                        for side in ('LEFT', 'RIGHT', 'LRSUM'):
                            shape = (H, W) if grey else (H, W, 3)
                            Image.fromarray(rng.randint(0, 256, shape).astype(np.uint8)).save(os.path.join(base, 'IMG', side, stem))
                        info = {
                            'valid': not (cam == 2 and view == 1 and light == 1 and name == 's02_m'),
                            'object': name, 'gender': name[-1], 'camidx': cam, 'lightidx': light,
                            'expression': expressions[(count // 2) % 3], 'position': 'forward' if view == 0 else 'backward',
                            'direction': ['left', 'right', 'front', 'upper'][count % 4],
                        }
                        paths = {
                            'root': name, 'left': 'IMG/LEFT/' + stem, 'right': 'IMG/RIGHT/' + stem, 'lrsum': 'IMG/LRSUM/' + stem,
                            'depth': 'DEPTH/DEPTH_%d_%d.npy' % (cam, view), 'normal': 'NORMAL/NORMAL_%d_%d.npy' % (cam, view),
                            'albedo': 'ALBEDO/ALBEDO_%d_%d.npy' % (cam, view), 'calibration': 'CALIBRATION',
                        }
                        if mask_file:
                            paths['mask'] = 'MASK/MASK_%d_%d.npy' % (cam, view)
                        intrinsic = [7000.0 + cam, 7010.0 + view, 0.0, W / 2 + 0.25 * cam, H / 2 - 0.5 * view, 0.01, -0.02, 0.0, 0.0]
                        pose = rng.normal(size=12)
                        params = {'intrinsic': _array_repr(intrinsic), 'pose': _array_repr(pose),
                                  'Lvalue': None if light == 1 else _array_repr(rng.normal(size=3)),
                                  'abvalue': _array_repr([1.0, 2.0])}
                        with open(os.path.join(base, 'JSON', 'INFO_%d_%d_%d.json' % (cam, view, light)), 'w') as fh:
                            json.dump({'INFO': info, 'PATH': paths, 'PARAMS': params}, fh)
                        count += 1
    return root


DATASET_OPT = {
    'path': None, 'gender': ['w', 'm'], 'viewpoint': [1, 2, 6], 'light': [1], 'focal_length_mm': 135,
    'expression': ['neutral', 'frowning'], 'distance': ['forward', 'backward'], 'direction': ['left', 'right', 'front', 'upper'],
    'dp_conversion': 'given', 'rot_input': {'left': False, 'right': False, 'center': False}, 'flip_lr': True, 'select_view': [1, 2],
}

USE_ALL = dict(use_multi=False, use_dual_pixel=True, use_center_img=True, use_mask=True, use_disparity=True, use_depth=True,
               use_idepth=True, use_normal=True, use_albedo=False, use_conf=False, use_raw=True)

MULTI_VIEW = dict(use_dual_pixel=True, use_center_img=False, use_mask=True, use_disparity=False, use_depth=True, use_idepth=True,
                  use_normal=False, use_albedo=False, use_conf=False)


def _crop(method, kind, ratio=0.75, factor=16, ch=32, cw=48, min_inlier=0.3):
    return {'method': method, 'type': kind, 'hard_crop': {'crop_width': cw, 'crop_height': ch},
            'soft_crop': {'crop_ratio': ratio, 'crop_factor': factor}, 'min_inlier': min_inlier, 'max_trial': 5}


def _photo(brightness=False, gamma=False, contrast=False, light=False):
    return {'brightness': brightness, 'gamma': gamma, 'contrast': contrast, 'light': light}


# name -> (option dict without dataset path, training flag, dataset kwargs, seeds, number of samples taken in order)
CASES = {
    'train_soft_light': (dict(USE_ALL, augmentation=['crop_aug', 'photo_aug'], crop_aug=_crop('random_crop', 'soft_crop'),
                              photo_aug=_photo(light=True)), True, {}, (5, 6, 7), 3),
    'eval_center': (dict(USE_ALL, augmentation=['crop_aug', 'photo_aug'], crop_aug=_crop('center_crop', 'soft_crop', ratio=1.0),
                         photo_aug=_photo()), False, {}, (1, 2, 3), 2),
    'mask_crop_hard': (dict(USE_ALL, use_raw=False, use_albedo=True, augmentation=['crop_aug'],
                            crop_aug=_crop('mask_random_crop', 'hard_crop', min_inlier=0.55)), True, {}, (11, 12, 13), 3),
    'photometric': (dict(USE_ALL, use_raw=False, augmentation=['crop_aug', 'photo_aug'], crop_aug=_crop('random_crop', 'hard_crop'),
                         photo_aug=_photo(True, True, True, True)), True, {}, (21, 22, 23), 2),
    'multi_view': (dict(USE_ALL, use_multi=True, use_raw=False, augmentation=['crop_aug'], crop_aug=_crop('random_crop', 'soft_crop'),
                        multi_view=MULTI_VIEW), True, {}, (31, 32, 33), 2),
    'f64_depth_mask_file': (dict(USE_ALL, use_normal=False, augmentation=[], use_center_img=False), True,
                            dict(depth_dtype=np.float64, mask_file=True), (41, 42, 43), 2),
    'no_augmentation_key': (dict(USE_ALL, use_raw=False, use_idepth=False, use_depth=False, augmentation=['photo_aug'],
                                 photo_aug=_photo(light=True)), True, {}, (51, 52, 53), 1),
}


def make_option(case, dataset_path):
    body, training, _, _, _ = CASES[case]
    d = dict(body)
    d.update(dataset_name='FaceDP', dataset=dict(DATASET_OPT, path=str(dataset_path)), mode='train' if training else 'test')
    if 'multi_view' not in d:
        d['multi_view'] = MULTI_VIEW
    return Opt(d), training


def seed_all(seeds):
    import random
    import torch
    random.seed(seeds[0])
    np.random.seed(seeds[1])
    torch.manual_seed(seeds[2])


def digest(value):
    """Hash + shape/dtype record of a tensor / array / scalar entry of a sample dict (bit-exact comparison without storing it)."""
    import hashlib
    import torch
    if torch.is_tensor(value):
        value = value.detach().cpu().numpy()
    if isinstance(value, np.ndarray):
        arr = np.ascontiguousarray(value)
        return {'shape': list(arr.shape), 'dtype': str(arr.dtype), 'sha256': hashlib.sha256(arr.tobytes()).hexdigest()}
    if isinstance(value, (list, tuple)):
        return {'list': [int(v) for v in value]}
    return {'value': value}
