"""The alternative kernel paths behind the DPF_* switches (README "Environment switches") are live code: unaligned shapes, LDS images
that do not fit and A/B measurements reach them.  The switches are read once per process, so each set runs the relevant parity tests
in a child interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETS = [
    # lean-sampler deformable kernels off: the role-split region kernels take the model's shapes too
    ({'DPF_DCN_LEAN': '0'}, 'test_deform_conv and not full_size'),
    # first-generation deformable kernels (global-memory gathers: what geometries without a fitting LDS image fall back to)
    ({'DPF_DCN_V1': '1'}, 'test_deform_conv and not full_size'),
    # the deformable conv backward's gcol products (grad_input and grad_offset kernels) on the fp32 matrix instruction instead of the f16 components
    ({'DPF_DCN_GCOL16': '0'}, 'test_deform_conv and not full_size'),
    # the 35-channel forward on the wider x halo (one workgroup per CU)
    ({'DPF_DCN_LEAN_WIDE12': '1'}, 'test_deform_conv and not full_size'),
    # consecutive-row weight-gradient tiles for dilated layers, one output plane per forward tile
    ({'DPF_W2_RSTEP': '0', 'DPF_G2_PZ': '1'}, 'test_conv_forward_backward'),
    # four output planes per tile wherever the geometry allows it; run-time column stride and the fp32 matrix instruction in the weight gradient
    ({'DPF_G2_PZ': '4', 'DPF_W2_SW1': '0', 'DPF_F32_X9': '0'}, 'test_conv_forward_backward'),
    # fp32 products from six bf16 partial products of exact three-way splits (round 5's default; the default is now three f16 products)
    ({'DPF_F32_X9': '1'}, 'test_conv_forward_backward or test_conv_epilogue_batchnorm_statistics or test_deform_conv'),
    # stride-1 convolutions on the exact-f32 matrix instruction instead of the bf16 partial products
    ({'DPF_IGEMM3': '0'}, 'test_conv_forward_backward or test_conv_epilogue_batchnorm_statistics'),
    # x9 convolutions: split as a phase of its own (one weight buffer) instead of in the MFMAs' shadow; forced chunk layouts
    ({'DPF_IGEMM3_SH': '0'}, 'test_conv_forward_backward or test_conv_epilogue_batchnorm_statistics or test_conv_f32_matrix_paths_agree or test_conv_f16_component_path_in_block_dynamic_range'),
    ({'DPF_IGEMM3_CC': '4'}, 'test_conv_forward_backward or test_conv_f32_matrix_paths_agree or test_conv_f16_component_path_in_block_dynamic_range'),
    # dilated 2-D layers on consecutive-row tiles (or, where that patch is too large, on igemm2) instead of the rows of one dilation phase
    ({'DPF_IGEMM3_RSTEP': '0'}, 'test_conv_forward_backward or test_conv_f32_matrix_paths_agree'),
    ({'DPF_IGEMM3_CC': '8'}, 'test_conv_forward_backward or test_conv_f32_matrix_paths_agree or test_conv_f16_component_path_in_block_dynamic_range'),
    # 4-byte stores in the conv tile epilogue (what outputs with W % 4 != 0 or an unaligned base get), bf16 operands on igemm2's own kernel
    ({'DPF_G2_VEC_STORE': '0'}, 'test_conv_forward_backward or test_conv_epilogue_batchnorm_statistics'),
    ({'DPF_IGEMM3_BF': '0'}, 'test_conv_operands_bf16 or test_conv2d_bf16_operands'),
    # deterministic mode (dpf_set_deterministic): one committing workgroup per address, phased tiles, integer accumulation -- same answers
    ({'DPF_DETERMINISTIC': '1'}, 'test_deform_conv or test_softargmin or test_batchnorm or test_depthwise or test_losses_against or '
                                 'test_conv_forward_backward or test_sync_batchnorm or test_norm_act_concat or test_conv_transpose3d'),
    # first-generation dense conv kernels (what unaligned shapes fall back to)
    ({'DPF_IGEMM2': '0', 'DPF_WGRAD2': '0', 'DPF_IGEMM2_TR2': '0'}, 'test_conv_forward_backward'),
]


# whole-model steps against the reference fixtures: on ONE stream (weight gradients in line, feature passes one after the other), and on the
# two element-exact fp32 matrix paths
E2E_SETS = [
    ({'DPF_FEATURES_TWO_STREAMS': '0', 'DPF_WGRAD_ASYNC': '0'}, 'test_gradients_and_adam_step_vs_reference_fixture or test_train_step_with_flat_grad_reducer_single_rank'),
    # the element-exact fp32 paths, whole model against the imported reference's fixtures: six bf16 partial products of exact
    # three-way splits (path 1), and v_mfma_f32_32x32x2_f32 (path 0); the c2 configuration (batch 4, 512 x 768) rides along
    ({'DPF_F32_X9': '1'}, 'test_gradients_and_adam_step_vs_reference_fixture or test_train_forward_stages_and_losses or test_c2_batch4'),
    # (path 0 accumulates every product in ONE sequential fp32 chain per output -- 864 v_mfma_f32_32x32x2_f32 steps for a 3x3x3 layer of 32
    # channels -- where paths 1 / 2 add exact 16-term block sums: its gradients sit FARTHER from the reference's than the default path's, up to
    # 5.4 x the reference's own thread-count noise on the normal head's last conv at 32 x 48 against <= 2.9 x on the default path; budget 8 x)
    ({'DPF_F32_X9': '0', 'DPF_TEST_K_SPREAD': '8'}, 'test_gradients_and_adam_step_vs_reference_fixture or test_train_forward_stages_and_losses or test_c2_batch4'),
]


@pytest.mark.gpu
@pytest.mark.parametrize('env_set,select', E2E_SETS, ids=['+'.join('%s=%s' % kv for kv in s[0].items()) for s in E2E_SETS])
def test_whole_model_parity_on_the_alternative_paths(env_set, select):
    env = dict(os.environ)
    env.update(env_set)
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_e2e.py'), '-m', 'gpu', '-x', '-q', '-k', select, '-p', 'no:cacheprovider']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    tail = (r.stdout or '')[-1500:] + (r.stderr or '')[-500:]
    assert r.returncode == 0, tail
    assert ' passed' in r.stdout and 'failed' not in r.stdout, tail


@pytest.mark.gpu
@pytest.mark.parametrize('env_set,select', SETS, ids=['+'.join('%s=%s' % kv for kv in s[0].items()) for s in SETS])
def test_parity_on_the_alternative_paths(env_set, select):
    env = dict(os.environ)
    env.update(env_set)
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_ops.py'), '-m', 'gpu', '-x', '-q', '-k', select, '-p', 'no:cacheprovider']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout or '')[-1500:] + (r.stderr or '')[-500:]
    assert r.returncode == 0, tail
    assert ' passed' in r.stdout and 'failed' not in r.stdout, tail
