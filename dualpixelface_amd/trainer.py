"""A small native trainer behind the reference's entry point.

The reference drives its plugin with ``pytorch_lightning.Trainer`` 1.4.9 (main.py:43-58: fit / test, ``ModelCheckpoint`` every epoch
into the workspace, ``resume_from_checkpoint``, validation every epoch, ``max_epochs = option.epoch``, DDP + SyncBatchNorm under
``accelerator == 'ddp'``).  pytorch_lightning is not part of this image, and the MI355X step keeps its gradients and Adam moments in
flat arenas that a generic trainer would not know how to checkpoint, so this module provides those behaviours natively:

  * ``Trainer(option).fit(model)`` / ``.test(model)`` -- epochs over ``model.train_dataloader()`` (or loaders passed in) calling the
    fused ``model.train_step`` (forward + loss + backward + bucketed RCCL all-reduce + Adam), the scheduler of
    ``scheduler_selector`` (StepLR(35, 0.5) / ExponentialLR(0.5) / CosineAnnealingLR(500, 1e-6) stepped per epoch), validation
    with the metric hooks every epoch, one checkpoint per epoch ``checkpoint_epoch=XX.ckpt``.
  * checkpoints hold ``state_dict`` under the reference's parameter names (a PL checkpoint of the reference loads into this
    model, and the other way round), the flat Adam moments, the step / epoch counters and the learning rate.
  * one process per GPU under torchrun: the training set is sharded by SAMPLE with torch's DistributedSampler (own generator
    seeded by (seed, epoch), padded so every rank runs the same number of equal-shape steps -- what PL does for the reference), gradient all-reduce through
    ``distributed.make_reducer``, SyncBatchNorm when ``option.sync_batch``; rank 0 writes checkpoints and logs.
"""
import json
import math
import os
import re
import time

import torch

from . import distributed as dd


def epoch_lr(option, epoch):
    """Learning rate of ``epoch`` (0-based) under the reference's schedulers (model_selector.py:45-58), stepped once per epoch."""
    lr0 = float(option.init_lr)
    name = getattr(option, 'scheduler', 'none')
    if name == 'steplr':
        return lr0 * 0.5 ** (epoch // 35)
    if name == 'explr':
        return lr0 * 0.5 ** epoch
    if name == 'cosanneal':
        return 1e-6 + (lr0 - 1e-6) * (1 + math.cos(math.pi * epoch / 500)) / 2
    if name == 'none':
        return lr0
    raise NotImplementedError('scheduler is not defined, please check your scheduler configuration !')


CKPT_VERSION = 2


class Trainer(object):
    def __init__(self, option, workspace_path=None, log_every=10, max_steps=None, rank=None, world_size=None):
        self.option = option
        self.workspace_path = str(workspace_path or getattr(option, 'workspace_path', '.'))
        self.log_every = log_every
        self.max_steps = max_steps              # optional cap on optimizer steps (smoke runs)
        if rank is None:
            rank, world_size, _ = dd.init_from_env()
        self.rank, self.world_size = rank, world_size
        self.epoch = 0
        self.global_step = 0
        self.history = []

    # ------------------------------------------------------------------ checkpoints
    def checkpoint_path(self, epoch):
        return os.path.join(self.workspace_path, 'checkpoint_epoch=%02d.ckpt' % epoch)

    def save_checkpoint(self, model, path=None):
        path = path or self.checkpoint_path(self.epoch)
        adam = model._adam or {}
        ckpt = {
            # PL 1.4.9 dump_checkpoint: the NEXT epoch to run and global_step + 1; 'dpf_ckpt_version' tells load_checkpoint which
            # convention a file follows (absent + no 'pytorch-lightning_version' = round-1 files: 'epoch' was the FINISHED epoch)
            'dpf_ckpt_version': CKPT_VERSION, 'epoch': self.epoch + 1, 'global_step': self.global_step + 1,
            'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()},
            'optimizer_states': [{'kind': 'flat_adam', 'step': int(adam.get('step', 0)),
                                  'm': adam['m'].detach().cpu() if 'm' in adam else None,
                                  'v': adam['v'].detach().cpu() if 'v' in adam else None}],
            'lr': epoch_lr(self.option, self.epoch),
            'hyper_parameters': {'model_name': getattr(self.option, 'model_name', 'stereodpnet')},
        }
        if self.rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = path + '.tmp'
            torch.save(ckpt, tmp)
            os.replace(tmp, path)
        return path

    def load_checkpoint(self, model, path, resume=True):
        """``resume``: restore optimizer moments and counters too (PL's resume_from_checkpoint); else weights only."""
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        weights = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt['model']
        epoch, global_step = self.epoch, self.global_step
        if resume:
            # Resolve the counters BEFORE anything is restored: a checkpoint whose convention cannot be determined must leave the model
            # and the trainer untouched (the error used to fire after load_state_dict and the counter update).
            epoch = int(ckpt.get('epoch', 0))                                 # PL restores current_epoch = ckpt['epoch']
            global_step = int(ckpt.get('global_step', 0))
            if int(ckpt.get('dpf_ckpt_version', 0)) >= 2:
                global_step = max(global_step - 1, 0)                         # our own files: undo dump_checkpoint's + 1 exactly
            elif 'pytorch-lightning_version' not in ckpt and 'epoch' in ckpt:
                # Unversioned files of this trainer exist in two generations that no key tells apart: round 1 stored the FINISHED epoch,
                # later revisions the NEXT epoch to run (and the plain global_step).  save_checkpoint names the file after the finished
                # epoch (checkpoint_epoch=NN.ckpt), so the name decides; without it the caller has to say (option.legacy_ckpt_epoch =
                # 'finished' | 'next') -- never guess: a wrong guess silently skips or repeats an epoch and shifts the LR schedule.
                m = re.search(r'checkpoint_epoch=(\d+)\.ckpt$', os.path.basename(path))
                conv = getattr(self.option, 'legacy_ckpt_epoch', None)
                if conv is None and m is not None:
                    if epoch == int(m.group(1)):
                        conv = 'finished'
                    elif epoch == int(m.group(1)) + 1:
                        conv = 'next'
                if conv not in ('finished', 'next'):
                    raise ValueError("unversioned checkpoint %r: cannot tell whether its 'epoch' = %d is the finished epoch or the next one to "
                                     "run; set option.legacy_ckpt_epoch to 'finished' or 'next', or load the weights only "
                                     "(load_checkpoint(..., resume=False) / --load_model without resume).  Nothing was restored." % (path, epoch))
                if conv == 'finished':
                    epoch += 1
        model.load_state_dict(weights, strict=bool(getattr(self.option, 'load_strict', True)))
        if resume:
            self.epoch, self.global_step = epoch, global_step
            states = ckpt.get('optimizer_states') or []
            if states and states[0].get('kind') == 'flat_adam' and states[0].get('m') is not None:
                dev = model.flat_parameters().device
                model._adam = {'m': states[0]['m'].to(dev), 'v': states[0]['v'].to(dev), 'step': int(states[0]['step'])}
        return ckpt

    # ------------------------------------------------------------------ loops
    def _shard(self, loader, epoch):
        """This rank's batches of one epoch.  world_size 1: the loader as it is.  Otherwise the loader is rebuilt over the same
        dataset with a DistributedSampler (shuffled by its own generator seeded with (seed, epoch), padded to a multiple of the
        world size): every rank sees the same number of batches with the same shapes, so the per-step collectives always pair up."""
        if self.world_size == 1:
            if hasattr(loader, 'set_epoch'):
                loader.set_epoch(epoch)                              # FaceDPBatcher: its own generator seeded with (seed, epoch)
            return loader
        key = id(loader)
        if getattr(self, '_dist_key', None) != key:
            from torch.utils.data import DataLoader
            from torch.utils.data.distributed import DistributedSampler
            self._dist_sampler = DistributedSampler(loader.dataset, num_replicas=self.world_size, rank=self.rank, shuffle=True,
                                                    seed=int(getattr(self.option, 'seed', 1)), drop_last=False)
            if hasattr(loader, 'with_sampler'):
                self._dist_loader = loader.with_sampler(self._dist_sampler)
            else:
                self._dist_loader = DataLoader(loader.dataset, batch_size=loader.batch_size, sampler=self._dist_sampler,
                                               num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                                               pin_memory=loader.pin_memory, drop_last=loader.drop_last)
            self._dist_key = key
        self._dist_sampler.set_epoch(epoch)
        return self._dist_loader

    def _to_device(self, batch, device):
        return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def _log(self, record):
        self.history.append(record)
        if self.rank == 0:
            os.makedirs(self.workspace_path, exist_ok=True)
            with open(os.path.join(self.workspace_path, 'log.jsonl'), 'a') as fh:
                fh.write(json.dumps(record) + '\n')

    def validate(self, model, loader, test=False):
        device = model.flat_parameters().device
        model.eval()
        outputs = []
        with torch.no_grad():
            for i, batch in enumerate(loader):
                batch = self._to_device(batch, device)
                outputs.append(None if (model.test_step if test else model.validation_step)(batch, i) is None else i)
        (model.test_epoch_end if test else model.validation_epoch_end)(outputs)
        rows = {n: f.get_value() for n, f in zip(model.metric_model.metric_name, model.metric_model.metric_func) if f.index > 0}
        for f in model.metric_model.metric_func:
            f.clear()
        model.train()
        return rows

    def fit(self, model, train_loader=None, val_loader=None):
        opt = self.option
        train_loader = train_loader if train_loader is not None else model.train_dataloader()
        if val_loader is None and hasattr(model, 'val_dataloader'):
            try:
                val_loader = model.val_dataloader()
            except Exception:
                val_loader = None
        if getattr(opt, 'load_model', None) and getattr(opt, 'mode', 'train') == 'train':
            self.load_checkpoint(model, opt.load_model, resume=bool(getattr(opt, 'load_strict', True)))
        dd.make_wait_group()                                 # while the ranks are in lock-step (see distributed.make_wait_group)
        dd.broadcast_flat(model.flat_parameters(), 0)
        reducer = dd.make_reducer(model) if self.world_size > 1 else None
        if self.world_size > 1 and getattr(opt, 'sync_batch', False):
            model.enable_sync_batchnorm()
        device = model.flat_parameters().device
        model.train()
        done = False
        while self.epoch < int(opt.epoch) and not done:
            lr = epoch_lr(opt, self.epoch)
            t0, n = time.time(), 0
            for batch in self._shard(train_loader, self.epoch):
                batch = self._to_device(batch, device)
                res = model.train_step(batch, reducer, lr=lr)
                self.global_step += 1
                n += int(next(iter(batch.values())).shape[0])
                if self.global_step % self.log_every == 0 or self.max_steps:
                    rec = {'epoch': self.epoch, 'step': self.global_step, 'lr': lr}
                    rec.update({k: float(v.detach()) for k, v in res.items() if 'loss' in k and torch.is_tensor(v)})
                    self._log(rec)
                if self.max_steps and self.global_step >= self.max_steps:
                    done = True
                    break
            if device.type == 'cuda':
                torch.cuda.synchronize()
            self._log({'epoch': self.epoch, 'epoch_seconds': time.time() - t0, 'samples_per_s_rank': n / max(time.time() - t0, 1e-9)})
            if val_loader is not None and self.rank == 0 and not done:
                rows = self.validate(model, val_loader)
                self._log({'epoch': self.epoch, 'metrics': rows})
            if self.world_size > 1:
                dd.wait_for_rank0()            # the other ranks wait here (host-side, own long timeout), not inside the next epoch's first all-reduce
            self.save_checkpoint(model)
            self.epoch += 1
        if reducer is not None:
            reducer.remove()
        return self.history

    def test(self, model, loader=None):
        loader = loader if loader is not None else model.test_dataloader()
        return self.validate(model, loader, test=True)
