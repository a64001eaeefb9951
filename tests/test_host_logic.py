"""CPU: host-side logic of the product (no GPU compute): ABI surface, sampler tables, state_dict contract, config."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from dualpixelface_amd import _lib, load_option
from dualpixelface_amd.sampler_tables import build_shift_tables, apply_tables_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_declares_and_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'dualpixelface_amd', 'csrc'), '-j8'])
    protos = _lib.parse_header()
    assert len(protos) >= 30
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), name
    nm = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in nm.splitlines() if ' T dpf_' in l}
    assert exported == set(protos), exported ^ set(protos)
    # plain-C boundary: no torch / ATen types in the header
    text = open(_lib.HEADER_PATH).read()
    assert 'at::' not in text and '#include <torch' not in text and 'Tensor' not in text


def test_ops_fail_loudly_without_gpu_tensors():
    from dualpixelface_amd import ops
    from dualpixelface_amd._lib import DpfError
    x = torch.randn(1, 4, 1, 8, 8)
    w = torch.randn(4, 4, 1, 3, 3)
    with pytest.raises(DpfError):
        ops.conv3d(x, w, None, 1, (0, 1, 1), 1)


def test_conv_operand_precision_context_nests_and_restores():
    """ops.conv_operands: the process-wide operand precision of the dense conv kernels follows the innermost context and is put back
    on exit, also when the body raises (host-only state of the library: no GPU needed)."""
    from dualpixelface_amd import ops
    from dualpixelface_amd._lib import lib
    get = lib().cdll.dpf_get_conv_operand_precision
    assert get() == 0 and not ops.CONV_OPERANDS_BF16
    with ops.conv_operands(True):
        assert get() == 1 and ops.CONV_OPERANDS_BF16
        with ops.conv_operands(False):                       # e.g. an fp32 node's backward running inside a bf16 forward
            assert get() == 0 and not ops.CONV_OPERANDS_BF16
        assert get() == 1 and ops.CONV_OPERANDS_BF16
        with ops.conv_operands(True):
            assert get() == 1
        assert get() == 1
    assert get() == 0 and not ops.CONV_OPERANDS_BF16
    with pytest.raises(RuntimeError):
        with ops.conv_operands(True):
            raise RuntimeError('body failed')
    assert get() == 0 and not ops.CONV_OPERANDS_BF16


def test_shift_tables_reproduce_reference_sampling(golden_dir):
    g = np.load(golden_dir + '/e2e_train_32x48_b2.npz')
    for fea_k, pre, d in (('fea_ref', 'shift_fwd_', -1.0), ('fea_tar', 'shift_bwd_', +1.0)):
        fea = torch.from_numpy(g[fea_k])
        out = apply_tables_reference(fea, build_shift_tables(fea.shape[2], fea.shape[3], d))
        for j, nm in enumerate(('nearest', 'bilinear', 'phase')):
            assert (out[:, :, j] - torch.from_numpy(g[pre + nm])).abs().max() < 1e-6, pre + nm
    # the nearest branch zero-fills the last column
    iy, wy, ix, wx, iy_inv, ix_inv = build_shift_tables(16, 24, -1.0)
    assert ix[0, 0, -1] == -1 and (ix[0, 0, :-1] >= 0).all()
    # inverse tables (deterministic gather adjoint): source coordinate -> output coordinate, consistent with the forward taps
    for fwd, inv in ((iy, iy_inv), (ix, ix_inv)):
        for m in range(3):
            for a in range(2):
                for out_pos, src_pos in enumerate(fwd[m, a].tolist()):
                    if src_pos >= 0:
                        assert out_pos in inv[m, a, :, src_pos].tolist()
                assert int((inv[m, a] >= 0).sum()) == int((fwd[m, a] >= 0).sum())
    # a fractional delta leaves the phase slot without taps (dpf_phase_shift fills it) ...
    iyf = build_shift_tables(16, 24, 0.5)[0]
    assert (iyf[2] == -1).all() and (iyf[1, 0] >= 0).any()


def test_phase_tables_reproduce_reference_fractional_shift(golden_dir):
    """... from tables that must reproduce the reference's irfft(onesided=False) arithmetic: row circulant + rank-one Hilbert term,
    evaluated densely here against outputs of the reference's subpixel_shift (tests/golden/shift_fractional.npz)."""
    from dualpixelface_amd.sampler_tables import build_phase_tables, is_fractional
    g = np.load(golden_dir + '/shift_fractional.npz')
    for ci in range(3):
        fea = torch.from_numpy(g['fea%d' % ci]).double()
        B, C, h, w = fea.shape
        y, x = torch.arange(h), torch.arange(w)
        sg = (1.0 - 2.0 * (y % 2)).double()
        for di, delta in enumerate(g['deltas']):
            for direction, sign in (('forward', 1.0), ('backward', -1.0)):
                d = sign * float(delta)
                if not is_fractional(d):
                    continue
                mr, hm, scale, mr_t, hm_t = build_phase_tables(h, w, d)
                Cm = mr.double()[(y[:, None] - y[None, :]) % h]
                Hm = hm.double()[(x[:, None] - x[None, :]) % w]
                S = torch.einsum('z,bczx->bcx', sg, fea)
                out = torch.einsum('yz,bczx->bcyx', Cm, fea) + scale * sg.view(1, 1, h, 1) * torch.einsum('xz,bcz->bcx', Hm, S).unsqueeze(2)
                want = torch.from_numpy(g['c%d_d%d_%s_phase' % (ci, di, direction)]).double()
                assert (out - want).abs().max() < 3e-6 * max(1.0, want.abs().max().item()), (ci, delta, direction)
                # adjoint tables are the index-reversed kernels
                assert torch.equal(mr_t, mr[(-y) % h]) and torch.equal(hm_t, hm[(-x) % w])
    with pytest.raises(NotImplementedError):
        build_phase_tables(15, 20, 0.5)


def test_state_dict_contract_and_flat_arena(golden_dir):
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import fill_by_recipe
    model = STEREODPNET(load_option())
    ref = json.load(open(golden_dir + '/state_dict_keys.json'))
    sd = model.state_dict()
    assert set(sd) == set(ref) and len(sd) == 511
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    flat = model.flat_parameters()
    assert flat.numel() == 3670492
    fill_by_recipe(model)
    p = dict(model.named_parameters())['aggregation.dres2.conv1.0.0.weight']
    assert p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + 4 * flat.numel()
    # alias keys of the doubly registered InstanceNorm are the same storage
    assert sd['cost_volume.attention_layer.normalize.weight'].data_ptr() == sd['cost_volume.attention_layer.mask_convs.3.1.weight'].data_ptr()
    # round trip + arena survives a dtype/device move
    other = STEREODPNET(load_option())
    other.load_state_dict(sd, strict=True)
    other.float()
    assert torch.equal(other.flat_parameters(), flat)
    opt, sched = model.configure_optimizers()
    assert opt[0].defaults['eps'] == 1e-5 and len(sched) == 1


def test_reference_style_plugin_entry():
    from runpy import run_path
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ns = run_path(os.path.join('src', 'model', 'stereodpnet', 'mainmodel.py'))
        assert 'STEREODPNET' in ns
        ls = run_path(os.path.join('src', 'loss', 'depth', 'smoothL1.py'))
        assert 'SMOOTHL1Loss' in ls
        assert 'COSINELoss' in run_path(os.path.join('src', 'loss', 'normal', 'cosine.py'))
    finally:
        os.chdir(cwd)


def test_reducer_stage_names_follow_the_bucket_cut():
    """The staged gradient exchange fires NAMED stages; a name maps to a bucket only when that bucket holds exactly that part of the network
    (hard-coded bucket numbers raised IndexError / exchanged half-built buckets with other cuts)."""
    from dualpixelface_amd import load_option
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.distributed import make_reducer, FlatGradReducer, stage_names
    model = STEREODPNET(load_option())
    r3 = make_reducer(model, nbuckets=3)
    assert r3.stage_of == {'aggregation': 1, 'normal': 2} and len(r3.buckets) == 3
    r3.remove()
    r2 = make_reducer(model, nbuckets=2)
    assert r2.stage_of == {} and len(r2.buckets) == 2             # aggregation and normal head share a bucket: nothing is staged early
    r2.stage_begin()
    r2.stage_launch(None); r2.stage_launch(7); r2.stage_launch(-1)  # refused, no IndexError
    assert r2.log == []
    r2.remove()
    r1 = make_reducer(model, nbuckets=1)
    assert r1.stage_of == {} and len(r1.buckets) == 1
    r1.remove()
    # a cut INSIDE the feature extractor: bucket 1 would mix feature and aggregation parameters -> unnamed
    pd = dict(model.named_parameters())
    layout = [(pd[name], off, numel) for name, off, numel, _ in model._layout]
    mid = layout[10][1]
    rc = FlatGradReducer(model.flat_gradients(zero=True), layout, [mid])
    assert stage_names(model, rc) == {}
    rc.remove()


def test_every_environment_switch_is_documented():
    """Sprawl guard: every DPF_* environment variable the product reads -- getenv / env_int / env_flag in the kernels' host
    code, os.environ in the Python package and bench.py -- is listed in README.md's "Environment switches" paragraph."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for f in glob.glob(os.path.join(root, 'dualpixelface_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'dualpixelface_amd', 'csrc', '*.h')):
        found |= set(re.findall(r'(?:getenv|env_int|env_flag)\(\s*"(DPF_[A-Z0-9_]+)"', open(f).read()))
    for f in glob.glob(os.path.join(root, 'dualpixelface_amd', '*.py')) + [os.path.join(root, 'bench.py')]:
        found |= set(re.findall(r"environ(?:\.get)?[\(\[]\s*'(DPF_[A-Z0-9_]+)'", open(f).read()))
    readme = open(os.path.join(root, 'README.md')).read()
    switches = readme[readme.index('Environment switches'):]
    # (DPF_ONE_DEVICE is a test hook of tests/test_gpu_distributed.py, documented there and in bench.py)
    missing = sorted(v for v in found if v not in switches and v != 'DPF_ONE_DEVICE')
    assert not missing, missing
    assert len(found) <= 40, (len(found), 'collapse switches that have a measured winner')
