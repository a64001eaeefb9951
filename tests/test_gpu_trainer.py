"""Native trainer on the GPU (``-m gpu``): checkpoint -> resume continues exactly like the uninterrupted run, the entry point
runs a synthetic smoke epoch, validation uses the metric hooks."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(opt):
    from dualpixelface_amd.plugin import STEREODPNET
    from dualpixelface_amd.recipe import fill_by_recipe
    m = STEREODPNET(opt)
    fill_by_recipe(m)
    return m.to('cuda').train()


def test_resume_matches_uninterrupted_training(tmp_path):
    from dualpixelface_amd import load_option
    from dualpixelface_amd.synthetic_data import synthetic_loader
    from dualpixelface_amd.trainer import Trainer
    opt = load_option()
    opt.epoch, opt.init_lr, opt.scheduler = 2, 1e-3, 'explr'
    loader = synthetic_loader(4, 32, 48, batch_size=2, seed=3)
    val = synthetic_loader(1, 32, 48, batch_size=1, seed=4)
    a = _model(opt)
    ta = Trainer(opt, str(tmp_path / 'a'), rank=0, world_size=1)
    hist = ta.fit(a, loader, val)
    assert ta.global_step == 4 and os.path.exists(ta.checkpoint_path(0)) and os.path.exists(ta.checkpoint_path(1))
    assert any('metrics' in h and set(h['metrics']) == {'absolute_dp', 'affine_dp', 'normal_dp'} for h in hist)
    # a second model resumes after epoch 0 and must land on the same parameters, moments and running statistics
    opt.load_model = ta.checkpoint_path(0)
    b = _model(opt)
    with torch.no_grad():
        b.flat_parameters().mul_(0.5)                      # the checkpoint must overwrite this
    tb = Trainer(opt, str(tmp_path / 'b'), rank=0, world_size=1)
    tb.fit(b, loader, None)
    assert tb.epoch == 2 and tb.global_step == 4
    pa, pb = a.flat_parameters().cpu(), b.flat_parameters().cpu()
    # not bit-identical: the atomics of the weight / input gradients reorder fp32 sums, and Adam turns a changed last bit of a
    # tiny gradient into a step of up to lr (5e-4 in epoch 1) -- two uninterrupted runs differ the same way.  Almost every
    # parameter must agree to 1e-5 and none may be off by more than the two steps taken since the checkpoint.
    diff = (pa - pb).abs()
    # measured over repeated runs (tools/flaky_margins.py): fraction 3e-6 .. 3.5e-4, max 1.5e-5 .. 1.9e-4
    assert (diff > 1e-5).float().mean().item() <= 5e-3 and diff.max().item() <= 1.1e-3, ((diff > 1e-5).float().mean().item(), diff.max().item())
    assert a._adam['step'] == b._adam['step'] == 4
    sa, sb = a.state_dict(), b.state_dict()
    k = 'feature_extraction.firstconv.0.1.running_var'
    assert torch.allclose(sa[k].cpu(), sb[k].cpu(), rtol=1e-4)
    assert int(sa['feature_extraction.firstconv.0.1.num_batches_tracked']) == int(sb['feature_extraction.firstconv.0.1.num_batches_tracked'])


def test_entry_point_smoke():
    """python main.py --config train_faceDP --workspace <name> with synthetic samples: two optimizer steps, a checkpoint and a log."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, 'main.py', '--config', 'train_faceDP', '--workspace', 'pytest_smoke', '--synthetic', '8', '--height', '64',
           '--width', '96', '--max_steps', '2']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ws = os.path.join(ROOT, 'workspace', 'stereodpnet', 'pytest_smoke')
    assert os.path.exists(os.path.join(ws, 'checkpoint_epoch=00.ckpt')) and os.path.exists(os.path.join(ws, 'log.jsonl'))
    ck = torch.load(os.path.join(ws, 'checkpoint_epoch=00.ckpt'), map_location='cpu', weights_only=False)
    # PL 1.4.9 dump_checkpoint convention: global_step + 1 (two optimizer steps -> 3), flagged by 'dpf_ckpt_version'
    assert ck['global_step'] == 3 and ck['dpf_ckpt_version'] == 2
    assert len(ck['state_dict']) == 512      # the reference's 511 keys + the lazy normal_estimator.grid (Q9)


def test_training_from_a_facedp_dataset_on_disk(tmp_path, monkeypatch):
    """The plugin's train_dataloader() over a FaceDP-format dataset on disk: loader_selector -> FaceDPLoader -> FaceDPBatcher
    (decode threads + device preprocessing) -> Trainer.fit; validation through val_dataloader()."""
    from dualpixelface_amd import load_option
    from dualpixelface_amd.facedp import FaceDPBatcher
    from dualpixelface_amd.trainer import Trainer
    from tests import facedp_fixture as fx
    data = fx.build_dataset(tmp_path / 'data', seed=0)
    monkeypatch.chdir(tmp_path)                                     # the index cache is written to the working directory
    opt = load_option()
    assert opt.crop_aug.soft_crop.crop_factor == 96 and opt.photo_aug.light is True and opt.use_raw is True
    opt.dataset.path = data
    opt.dataset.viewpoint = [1, 2, 6]
    opt.crop_aug.soft_crop.crop_factor = 16                          # 48 x 64 frames -> 32 x 48 crops
    opt.use_raw, opt.workers, opt.batch_size, opt.epoch = False, 2, 2, 1
    m = _model(opt)
    loader = m.train_dataloader()
    assert isinstance(loader, FaceDPBatcher) and len(loader.dataset) == 7 and os.path.exists('FaceDP_train_single.npy')
    val = m.val_dataloader()
    assert val.batch_size == 1 and len(val.dataset) == 4
    tr = Trainer(opt, str(tmp_path / 'ws'), log_every=1, rank=0, world_size=1)
    p0 = m.flat_parameters().clone()
    hist = tr.fit(m, loader, val)
    assert tr.global_step == 4 and torch.isfinite(m.flat_parameters()).all() and not torch.equal(p0, m.flat_parameters())
    losses = [v for h in hist for k, v in h.items() if 'loss' in k]
    assert losses and all(l == l for l in losses)
    assert any('metrics' in h for h in hist)


def test_reference_entry_point_call_sequence_through_the_pl_shim(tmp_path, monkeypatch):
    """compat/pytorch_lightning: the exact sequence of calls the reference's main.py makes (main.py:20-61 -- seed_everything,
    TensorBoardLogger, LearningRateMonitor, ModelCheckpoint, Trainer(<its 14 keyword arguments>).fit(model=model), then a resumed
    Trainer and .test) on the native trainer, over a FaceDP dataset on disk."""
    import sys
    from dualpixelface_amd import load_option
    from tests import facedp_fixture as fx
    monkeypatch.syspath_prepend(os.path.join(ROOT, 'compat'))
    for name in [n for n in sys.modules if n == 'pytorch_lightning' or n.startswith('pytorch_lightning.')]:
        monkeypatch.delitem(sys.modules, name)
    from pytorch_lightning import Trainer, seed_everything
    from pytorch_lightning import loggers as pl_loggers
    from pytorch_lightning.callbacks import LearningRateMonitor, ModelCheckpoint
    data = fx.build_dataset(tmp_path / 'data', seed=0)
    monkeypatch.chdir(tmp_path)
    opt = load_option()
    opt.dataset.path, opt.dataset.viewpoint = data, [1, 2, 6]
    opt.crop_aug.soft_crop.crop_factor = 16
    opt.use_raw, opt.workers, opt.batch_size, opt.epoch = False, 2, 2, 2
    opt.workspace_path, opt.logger_path = str(tmp_path / 'ws'), str(tmp_path / 'ws' / 'log')
    seed_everything(1)
    model = _model(opt)

    def make_trainer(resume):
        logger = pl_loggers.TensorBoardLogger(str(opt.logger_path))
        callbacks = [LearningRateMonitor(logging_interval='step'),
                     ModelCheckpoint(dirpath=str(opt.workspace_path), filename='checkpoint_{epoch:02d}', save_top_k=-1, period=1)]
        return Trainer(logger=logger, checkpoint_callback=True, callbacks=callbacks, resume_from_checkpoint=resume,
                       check_val_every_n_epoch=1, accelerator=opt.accelerator, benchmark=True, deterministic=False,
                       gpus=torch.cuda.device_count(), precision=opt.precision, max_epochs=opt.epoch, sync_batchnorm=opt.sync_batch,
                       amp_level='O2', profiler='pytorch')

    runner = make_trainer(None)
    runner.fit(model=model)
    ck0, ck1 = (os.path.join(opt.workspace_path, 'checkpoint_epoch=%02d.ckpt' % e) for e in (0, 1))
    assert os.path.exists(ck0) and os.path.exists(ck1) and runner.native.global_step == 8
    assert os.path.exists(os.path.join(opt.logger_path, 'scalars.jsonl'))
    # resume from the first epoch's checkpoint: one more epoch is run, not two
    opt.load_model = None
    model2 = _model(opt)
    runner2 = make_trainer(ck0)
    runner2.fit(model=model2)
    assert runner2.native.epoch == 2 and runner2.native.global_step == 8
    rows = make_trainer(None).test(model=model2, verbose=False)
    assert set(rows[0]) == {'absolute_dp', 'affine_dp', 'normal_dp'}


def test_main_py_train_checkpoint_then_test_on_a_facedp_dataset(tmp_path):
    """The reference's command-line flow (main.py:13-19, 60-70) end to end on a FaceDP-layout dataset on disk, with this repo's main.py:
    `--config <train> --workspace run` trains one epoch (FaceDPLoader -> FaceDPBatcher -> native trainer, validation on rank 0) and
    writes the per-epoch checkpoint; `--config <test> --workspace eval --load_model <ckpt>` loads it through model_selector (strict,
    reference key names) and runs the test split through the metric hooks: the metric table is printed.  The working directory is a
    config tree laid out like the reference's (config_/, src/model/<name>/config.json, dataloader/<dataset>/config.json,
    dataloader/preprocess/*.json) whose dataset path points at the fixture."""
    import json
    from tests import facedp_fixture as fx
    data = fx.build_dataset(tmp_path / 'data', seed=0)
    root = tmp_path / 'tree'
    (root / 'config_').mkdir(parents=True)
    (root / 'dataloader' / 'FaceDP').mkdir(parents=True)
    (root / 'dataloader' / 'preprocess').mkdir(parents=True)
    os.symlink(os.path.join(ROOT, 'src'), root / 'src')                       # plugin entry points: src/model/<name>/mainmodel.py
    ds = json.load(open(os.path.join(ROOT, 'dataloader', 'FaceDP', 'config.json')))
    ds.update(path=str(data), viewpoint=[1, 2, 6])
    json.dump(ds, open(root / 'dataloader' / 'FaceDP' / 'config.json', 'w'))
    for name in ('config_train', 'config_test'):
        pre = json.load(open(os.path.join(ROOT, 'dataloader', 'preprocess', name + '.json')))
        pre['crop_aug']['soft_crop']['crop_factor'] = 16                      # 48 x 64 frames -> 32 x 48 crops
        json.dump(pre, open(root / 'dataloader' / 'preprocess' / (name + '.json'), 'w'))
    for src_cfg, dst_cfg in (('train_faceDP', 'train_tiny'), ('eval_faceDP', 'test_tiny')):
        cfg = json.load(open(os.path.join(ROOT, 'config_', src_cfg + '.json')))
        cfg.update(epoch=1, batch_size=2, workers=2, use_raw=False, accelerator='dp')
        json.dump(cfg, open(root / 'config_' / (dst_cfg + '.json'), 'w'))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--config', 'train_tiny', '--workspace', 'run'], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ws = root / 'workspace' / 'stereodpnet' / 'run'
    ck = ws / 'checkpoint_epoch=00.ckpt'
    assert ck.exists()
    log = [json.loads(l) for l in open(ws / 'log.jsonl')]
    assert any('metrics' in rec for rec in log) and any('epoch_seconds' in rec for rec in log)      # validation ran on rank 0; the epoch finished
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--config', 'test_tiny', '--workspace', 'eval', '--load_model', str(ck)],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for name in ('absolute_dp', 'affine_dp', 'normal_dp'):
        assert name in r.stdout, r.stdout[-2000:]
