"""Stand-in for the handful of pytorch_lightning 1.4.9 names the reference's entry point uses (main.py:7-10,20-61), so that the
reference's OWN `main.py` drives the MI355X plugins without PyTorch Lightning (PL 1.4.9 does not install against torch 2.x):

    PYTHONPATH=<this repo>/compat:<this repo> python main.py --config train_faceDP --workspace base      # inside a reference checkout

`Trainer(...)` takes main.py's keyword arguments and maps `fit` / `test` onto the native trainer (dualpixelface_amd/trainer.py: flat
parameter arena, fused Adam, RCCL gradient all-reduce, SyncBatchNorm, per-epoch checkpoints in PL's `checkpoint_epoch=NN.ckpt` naming).
Put this directory on PYTHONPATH only where the real package is absent -- it shadows it.
"""
import os
import random

import numpy as np
import torch
import torch.nn as nn

from . import callbacks, loggers  # noqa: F401

__version__ = '1.4.9+dpf'


def seed_everything(seed=None, workers=False):
    seed = int(seed if seed is not None else 0)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PL_GLOBAL_SEED'] = str(seed)
    return seed


class LightningModule(nn.Module):
    """nn.Module with the LightningModule attributes the reference's model classes touch."""

    current_epoch = 0
    global_step = 0

    def save_hyperparameters(self, *args, **kwargs):
        pass

    def log(self, *args, **kwargs):
        pass

    def log_dict(self, *args, **kwargs):
        pass


class Trainer(object):
    """main.py:43-58.  Arguments with no meaning on this stack (benchmark, deterministic, amp_level, profiler, gpus -- one process per
    GPU comes from the launcher) are accepted and ignored."""

    def __init__(self, logger=None, checkpoint_callback=True, callbacks=None, resume_from_checkpoint=None, check_val_every_n_epoch=1,
                 accelerator=None, benchmark=None, deterministic=None, gpus=None, precision=32, max_epochs=None, sync_batchnorm=False,
                 amp_level=None, profiler=None, max_steps=None, log_every_n_steps=10, **ignored):
        self.logger = logger
        self.callbacks = list(callbacks or [])
        self.checkpoint_callback = bool(checkpoint_callback)
        self.resume_from_checkpoint = resume_from_checkpoint
        self.check_val_every_n_epoch = int(check_val_every_n_epoch or 1)
        self.accelerator, self.precision = accelerator, precision
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.sync_batchnorm = bool(sync_batchnorm)
        self.log_every_n_steps = int(log_every_n_steps)
        self.native = None

    def _native(self, model):
        from dualpixelface_amd.distributed import init_from_env
        from dualpixelface_amd.trainer import Trainer as NativeTrainer
        opt = model.option
        if self.max_epochs is not None:
            opt.epoch = int(self.max_epochs)
        opt.sync_batch = self.sync_batchnorm
        if str(self.precision) in ('16', 'bf16'):
            opt.precision = self.precision
        workspace = None
        for cb in self.callbacks:
            if isinstance(cb, callbacks.ModelCheckpoint) and cb.dirpath:
                workspace = cb.dirpath
        if workspace is None:
            workspace = getattr(opt, 'workspace_path', None) or (self.logger.save_dir if self.logger is not None else '.')
        rank, world, local = init_from_env()
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            model.to(torch.device('cuda', local))
        self.native = NativeTrainer(opt, str(workspace), log_every=self.log_every_n_steps, max_steps=self.max_steps, rank=rank,
                                    world_size=world)
        return self.native

    def fit(self, model, train_dataloader=None, val_dataloaders=None, **kw):
        native = self._native(model)
        if self.resume_from_checkpoint:
            model.option.load_model = str(self.resume_from_checkpoint)
        elif getattr(model.option, 'load_model', None) and not getattr(model.option, 'load_strict', True):
            native.load_checkpoint(model, model.option.load_model, resume=False)          # weights only (main.py:47)
            model.option.load_model = None
        history = native.fit(model, train_dataloader, val_dataloaders)
        if self.logger is not None:
            self.logger.log_history(history)
        return history

    def test(self, model, test_dataloaders=None, verbose=True, **kw):
        native = self._native(model)
        rows = native.test(model, test_dataloaders)
        if verbose:
            print(rows)
        return [rows]

    def validate(self, model, val_dataloaders=None, verbose=True, **kw):
        native = self._native(model)
        loader = val_dataloaders if val_dataloaders is not None else model.val_dataloader()
        rows = native.validate(model, loader)
        if verbose:
            print(rows)
        return [rows]
