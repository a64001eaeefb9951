"""FaceDP data path, CPU side: the product's index / readers / augmentation draws feeding the numpy oracle
must reproduce, bit for bit, the sample dicts the reference loader produced on the same seeded on-disk dataset
(tests/golden/facedp_samples.json, made by tests/golden/make_golden_facedp.py).  No GPU and no /root/reference needed."""
import json
import os

import numpy as np
import pytest
import torch

from dualpixelface_amd import facedp
from oracle import facedp_preprocess as oracle
from tests import facedp_fixture as fx

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'facedp_samples.json')))


@pytest.fixture(scope='module')
def datasets(tmp_path_factory):
    built = {}

    def get(case):
        kwargs = fx.CASES[case][2]
        key = json.dumps({k: str(v) for k, v in kwargs.items()}, sort_keys=True)
        if key not in built:
            built[key] = fx.build_dataset(tmp_path_factory.mktemp('facedp'), seed=0, **kwargs)
        return built[key]
    return get


def raw_arrays(raw):
    return {n: getattr(raw, n) for n in ('left', 'right', 'center', 'depth', 'file_mask', 'normal', 'albedo')}


def oracle_sample(ds, index):
    """Host half of the product (read + draw) + oracle arithmetic -> the reference's sample dict as numpy arrays."""
    raw, flags, refs = ds.read(index)
    plan = facedp.draw_plan(ds.opt, raw, flags)
    win = (plan.x0, plan.y0, plan.ch, plan.cw)
    shifts = {k: (v.numpy() if v is not None else None) for k, v in plan.shift.items()}
    sample = oracle.sample_view(raw_arrays(raw), flags, win, shifts, plan.lut, True, raw.a, raw.b)
    K = raw.K.copy()
    K[0, 2] -= plan.x0
    K[1, 2] -= plan.y0
    sample.update(K=K, P=raw.P, abvalue=raw.abvalue, metadata=raw.metadata, coords=np.asarray([plan.x0, plan.y0]))
    h, w = raw.image_shape()
    if ds.opt.use_raw:
        rv = oracle.sample_view(raw_arrays(raw), flags, (0, 0, h, w), None, None, False, raw.a, raw.b)
        sample.update({'raw_' + k: v for k, v in rv.items()})
    if ds.use_multi:
        views = [oracle.sample_view(raw_arrays(r), f, (0, 0) + r.image_shape(), None, None, False, r.a, r.b) for r, f in refs]
        for name in views[0]:
            sample[name + 's'] = np.concatenate([v[name] for v in views], axis=0)
        for name, attr in (('Ks', 'K'), ('Ps', 'P'), ('abvalues', 'abvalue'), ('metadatas', 'metadata')):
            sample[name] = np.asarray([getattr(r, attr) for r, _ in refs])
    sample.update(ds.names(index))
    return sample


@pytest.mark.parametrize('case', sorted(fx.CASES))
def test_host_pipeline_and_oracle_reproduce_the_reference_loader(case, datasets, tmp_path):
    option, training = fx.make_option(case, datasets(case))
    ds = facedp.FaceDPLoader(option, training, cache_dir=str(tmp_path))
    gold = GOLDEN[case]
    assert len(ds) == gold['length']
    assert [os.path.relpath(e['tar_view'], datasets(case)) for e in ds.pathdata] == gold['index']
    fx.seed_all(fx.CASES[case][3])
    for i, want in enumerate(gold['samples']):
        got = oracle_sample(ds, i)
        assert sorted(got) == sorted(want), (case, i)
        for key, rec in want.items():
            assert fx.digest(got[key]) == rec, (case, i, key)


def test_full_arrays_of_sample0(datasets, tmp_path):
    """Same check with the stored arrays (readable diff if a hash ever moves)."""
    case = 'train_soft_light'
    option, training = fx.make_option(case, datasets(case))
    ds = facedp.FaceDPLoader(option, training, cache_dir=str(tmp_path))
    fx.seed_all(fx.CASES[case][3])
    got = oracle_sample(ds, 0)
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'facedp_sample0.npz'))
    for key in gold.files:
        np.testing.assert_array_equal(np.asarray(got[key]), gold[key], err_msg=key)


def test_index_cache_round_trip_and_reference_format(datasets, tmp_path):
    case = 'multi_view'
    option, training = fx.make_option(case, datasets(case))
    first = facedp.FaceDPLoader(option, training, cache_dir=str(tmp_path))
    cache = tmp_path / 'FaceDP_train_multi.npy'                      # loader.py:93-102 naming
    assert cache.is_file()
    entries, n = np.load(str(cache), allow_pickle=True)              # the reference's own read (loader.py:110)
    assert n == len(first) and entries[0]['ref_view'] and set(entries[0]) == {'ref_view', 'tar_view', 'parentdir'}
    again = facedp.FaceDPLoader(option, training, cache_dir=str(tmp_path))
    assert again.pathdata == first.pathdata
    # one sub-view of s02_m is invalid: its slot is padded with the last valid view (path_reader.py:107-109)
    padded = [e for e in first.pathdata if len(set(e['ref_view'])) == 1]
    assert padded and all(len(e['ref_view']) == 2 for e in first.pathdata)


def test_filters(datasets, tmp_path):
    case = 'train_soft_light'
    option, training = fx.make_option(case, datasets(case))
    base = len(facedp.build_index(option, option.dataset.path, training))
    option.dataset.light = [1, 2]
    assert len(facedp.build_index(option, option.dataset.path, training)) == 2 * base + 1      # the invalid entry has light 1
    option.dataset.gender = ['m']
    option.dataset.viewpoint = [6]
    names = [os.path.basename(e['tar_view']) for e in facedp.build_index(option, option.dataset.path, training)]
    assert names and all(n.startswith('INFO_6_') for n in names)
    with pytest.raises(FileNotFoundError):
        facedp.read_split(str(tmp_path), True)


def test_array_literal_parser_is_not_eval():
    assert facedp.parse_array_literal('array([1., 2.5, -3e2])') == [1.0, 2.5, -300.0]
    assert facedp.parse_array_literal('array([[1, 2],\n       [3, 4]])') == [[1, 2], [3, 4]]
    with pytest.raises(Exception):
        facedp.parse_array_literal('array([__import__("os").system("true")])')
    assert facedp.parse_array_literal(None) is None


def test_crop_geometry():
    crop = fx.Opt({'method': 'center_crop', 'type': 'soft_crop', 'hard_crop': {'crop_width': 576, 'crop_height': 768},
                   'soft_crop': {'crop_ratio': 0.75, 'crop_factor': 96}, 'min_inlier': 0.3, 'max_trial': 5})
    assert facedp.crop_size((1024, 1536), crop) == (768, 1152)         # the shipped config_train.json on a FaceDP frame
    crop.soft_crop.crop_ratio = 1.0
    assert facedp.crop_size((1024, 1536), crop) == (960, 1536)         # config_test.json: multiples of 96 only
    assert facedp.draw_crop((1024, 1536), (960, 1536), crop) == (0, 32)
    crop.type = 'hard_crop'
    assert facedp.crop_size((1024, 1536), crop) == (768, 576)
    crop.method = 'bogus'
    with pytest.raises(NotImplementedError):
        facedp.draw_crop((10, 10), (5, 5), crop)


def test_photometric_lut_equals_pil_on_the_image():
    """The 256-entry table is the whole brightness -> gamma -> contrast chain of the PIL ops applied to the picture itself."""
    from PIL import Image, ImageEnhance
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (24, 40, 3)).astype(np.uint8)
    for br, ga, co in ((0.83, 0, 0), (0, 1.13, 0), (0, 0, 0.77), (1.17, 0.74, 1.09)):
        lut = facedp.photometric_lut(img, br, ga, co)
        pil = Image.fromarray(img)
        if br:
            pil = ImageEnhance.Brightness(pil).enhance(br)
        if ga:
            pil = pil.point([int(255 * pow(v / 255., ga)) for v in range(256)] * 3)
        if co:
            pil = ImageEnhance.Contrast(pil).enhance(co)
        want = np.asarray(pil)
        got = np.stack([lut[c][img[..., c]] for c in range(3)], axis=2)
        np.testing.assert_array_equal(got, want)


def test_batcher_order_is_independent_of_global_rng(datasets, tmp_path):
    case = 'train_soft_light'
    option, training = fx.make_option(case, datasets(case))
    ds = facedp.FaceDPLoader(option, training, cache_dir=str(tmp_path))
    b = facedp.FaceDPBatcher(ds, batch_size=2, shuffle=True, seed=1)
    b.set_epoch(3)
    first = b._order()
    torch.rand(5)
    assert b._order() == first and sorted(first) == list(range(len(ds)))
    b.set_epoch(4)
    assert b._order() != first
    assert len(b) == 4 and len(facedp.FaceDPBatcher(ds, 2, drop_last=True)) == 3


def test_device_half_fails_loudly_without_gpu(datasets, tmp_path):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    case = 'eval_center'
    option, training = fx.make_option(case, datasets(case))
    ds = facedp.FaceDPLoader(option, training, cache_dir=str(tmp_path))
    with pytest.raises(Exception):
        ds[0]
