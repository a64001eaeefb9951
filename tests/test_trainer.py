"""Trainer logic that needs no GPU: the reference's schedules and the checkpoint round trip on a stand-in model."""
import os

import pytest

import torch

from dualpixelface_amd.config import load_option
from dualpixelface_amd.trainer import Trainer, epoch_lr


def test_schedules_match_torch():
    opt = load_option()
    for name, mk in (('steplr', lambda o: torch.optim.lr_scheduler.StepLR(o, 35, 0.5)),
                     ('explr', lambda o: torch.optim.lr_scheduler.ExponentialLR(o, 0.5)),
                     ('cosanneal', lambda o: torch.optim.lr_scheduler.CosineAnnealingLR(o, 500, 1e-6))):
        opt.scheduler = name
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.SGD([p], lr=float(opt.init_lr))
        s = mk(o)
        for epoch in range(80):
            assert abs(o.param_groups[0]['lr'] - epoch_lr(opt, epoch)) <= 1e-12 + 1e-9 * float(opt.init_lr), (name, epoch)
            o.step()
            s.step()


class _Stub(torch.nn.Module):
    """flat-arena stand-in with the attributes the trainer touches"""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.arange(6, dtype=torch.float32))
        self.register_buffer('running', torch.ones(3))
        self._adam = None
        self.steps = []

    def flat_parameters(self):
        return self.w.data

    def train_step(self, batch, reducer=None, lr=None):
        if self._adam is None:
            self._adam = {'m': torch.zeros(6), 'v': torch.zeros(6), 'step': 0}
        self._adam['step'] += 1
        self._adam['m'] += batch['x'].mean()
        self.w.data -= lr * 1000 * batch['x'].mean()
        self.steps.append(lr)
        return {'final_loss': batch['x'].mean()}


def test_checkpoint_every_epoch_and_resume(tmp_path):
    opt = load_option()
    opt.epoch, opt.scheduler = 3, 'explr'
    data = [{'x': torch.full((2, 1), float(i))} for i in range(4)]
    m = _Stub()
    tr = Trainer(opt, str(tmp_path), rank=0, world_size=1)
    tr.fit(m, data, None)
    assert [os.path.exists(tr.checkpoint_path(e)) for e in range(3)] == [True] * 3
    assert m.steps[:4] == [1e-4] * 4 and m.steps[4:8] == [5e-5] * 4 and len(m.steps) == 12
    # resume from the epoch-1 checkpoint: counters, moments and weights come back, training continues with epoch 2
    m2 = _Stub()
    opt.load_model = tr.checkpoint_path(1)
    tr2 = Trainer(opt, str(tmp_path / 'resumed'), rank=0, world_size=1)
    tr2.fit(m2, data, None)
    assert tr2.epoch == 3 and tr2.global_step == 12 and len(m2.steps) == 4 and m2.steps[0] == 2.5e-5
    assert torch.equal(m2.w.data, m.w.data) and m2._adam['step'] == 12 and torch.equal(m2._adam['m'], m._adam['m'])
    ck = torch.load(tr.checkpoint_path(2), weights_only=False)
    assert set(ck['state_dict']) == {'w', 'running'} and ck['optimizer_states'][0]['kind'] == 'flat_adam'
    # file conventions: ours (versioned) stores PL's epoch + 1 / global_step + 1; a round-1 file (no version key, no PL key) stored the
    # FINISHED epoch and the plain step count; a PL file is taken as PL restores it
    assert ck['dpf_ckpt_version'] == 2 and ck['epoch'] == 3 and ck['global_step'] == 13
    ck1 = torch.load(tr.checkpoint_path(1), weights_only=False)
    legacy = dict(ck1, epoch=1, global_step=8)
    del legacy['dpf_ckpt_version']
    # the two unversioned generations are told apart by the file name save_checkpoint gives (NN = the finished epoch) -- never guessed
    os.makedirs(str(tmp_path / 'r1'))
    torch.save(legacy, str(tmp_path / 'r1' / 'checkpoint_epoch=01.ckpt'))             # round 1: 'epoch' = finished epoch
    tr3 = Trainer(opt, str(tmp_path / 'legacy'), rank=0, world_size=1)
    tr3.load_checkpoint(_Stub(), str(tmp_path / 'r1' / 'checkpoint_epoch=01.ckpt'))
    assert tr3.epoch == 2 and tr3.global_step == 8
    base = dict(legacy, epoch=2, global_step=8)                                       # later revisions: 'epoch' = next epoch, plain step
    os.makedirs(str(tmp_path / 'r2'))
    torch.save(base, str(tmp_path / 'r2' / 'checkpoint_epoch=01.ckpt'))
    tr5 = Trainer(opt, str(tmp_path / 'legacy2'), rank=0, world_size=1)
    tr5.load_checkpoint(_Stub(), str(tmp_path / 'r2' / 'checkpoint_epoch=01.ckpt'))
    assert tr5.epoch == 2 and tr5.global_step == 8
    torch.save(legacy, str(tmp_path / 'legacy.ckpt'))                                 # renamed file: refuse to guess ...
    tr6 = Trainer(opt, str(tmp_path / 'legacy3'), rank=0, world_size=1)
    fresh = _Stub()
    w_before = fresh.w.data.clone()
    with pytest.raises(ValueError):
        tr6.load_checkpoint(fresh, str(tmp_path / 'legacy.ckpt'))
    # ... and the refusal restores NOTHING: weights, optimizer state and the trainer's counters are as they were
    assert torch.equal(fresh.w.data, w_before) and tr6.epoch == 0 and tr6.global_step == 0 and not getattr(fresh, '_adam', None)
    tr6.load_checkpoint(fresh, str(tmp_path / 'legacy.ckpt'), resume=False)           # weights only always works
    assert torch.equal(fresh.w.data, legacy['state_dict']['w']) and tr6.epoch == 0
    opt.legacy_ckpt_epoch = 'finished'                                                # ... unless the caller says which
    tr6.load_checkpoint(_Stub(), str(tmp_path / 'legacy.ckpt'))
    assert tr6.epoch == 2
    opt.legacy_ckpt_epoch = 'next'
    tr6.load_checkpoint(_Stub(), str(tmp_path / 'legacy.ckpt'))
    assert tr6.epoch == 1
    del opt.legacy_ckpt_epoch
    pl = dict(legacy, epoch=2, global_step=9)
    pl['pytorch-lightning_version'] = '1.4.9'
    torch.save(pl, str(tmp_path / 'pl.ckpt'))
    tr4 = Trainer(opt, str(tmp_path / 'pl'), rank=0, world_size=1)
    tr4.load_checkpoint(_Stub(), str(tmp_path / 'pl.ckpt'))
    assert tr4.epoch == 2 and tr4.global_step == 9


def test_sample_sharding_equal_steps_on_every_rank():
    """world_size 3 over 7 samples, batch 2 (neither divides): every rank must see the same number of batches with the same shapes
    (DistributedSampler padding), a different permutation per epoch that does not depend on the global RNG, and all samples covered."""
    from torch.utils.data import DataLoader, Dataset

    class DS(Dataset):
        def __len__(self):
            return 7

        def __getitem__(self, i):
            return {'x': torch.tensor([float(i)])}

    opt = load_option()
    loader = DataLoader(DS(), batch_size=2, shuffle=True)
    per_epoch = []
    for epoch in range(2):
        torch.manual_seed(epoch * 17)          # the global RNG state must not matter
        seen = []
        for r in range(3):
            tr = Trainer(opt, '.', rank=r, world_size=3)
            seen.append([b['x'].flatten().tolist() for b in tr._shard(loader, epoch)])
        shapes = [[len(b) for b in s] for s in seen]
        assert shapes[0] == shapes[1] == shapes[2] and len(shapes[0]) == 2, shapes
        flat = sorted(int(v) for s in seen for b in s for v in b)
        assert set(flat) == set(range(7)) and len(flat) == 9          # padded to 3 x 3 samples
        per_epoch.append(seen)
    assert per_epoch[0] != per_epoch[1]
    assert list(Trainer(opt, '.', rank=0, world_size=1)._shard([1, 2, 3], 0)) == [1, 2, 3]


def test_sample_sharding_eight_ranks_uneven_last_batch():
    """BASELINE configs[2]'s world size: 8 ranks over 13 samples, batch 2 -- the sampler pads to 16 samples, every rank runs ONE batch of the
    same shape per epoch (so the per-step collectives pair up), every sample is seen."""
    from torch.utils.data import DataLoader, Dataset

    class DS(Dataset):
        def __len__(self):
            return 13

        def __getitem__(self, i):
            return {'x': torch.tensor([float(i)])}

    opt = load_option()
    loader = DataLoader(DS(), batch_size=2, shuffle=True)
    seen = []
    for r in range(8):
        tr = Trainer(opt, '.', rank=r, world_size=8)
        seen.append([b['x'].flatten().tolist() for b in tr._shard(loader, 3)])
    assert all(len(s) == 1 and len(s[0]) == 2 for s in seen), seen
    flat = sorted(int(v) for s in seen for b in s for v in b)
    assert set(flat) == set(range(13)) and len(flat) == 16


def test_two_rank_trainer_gloo(tmp_path):
    """Two gloo ranks through Trainer.fit on a batch count that is not divisible by the world size, with a short last batch and a
    rank-0-only validation pass: both ranks must finish every epoch with the same number of steps (no hang, no mixed epochs)."""
    import torch.multiprocessing as mp
    mp.spawn(_two_rank_worker, args=(2, str(tmp_path), 29500 + os.getpid() % 2000), nprocs=2, join=True)
    steps = [int(open(os.path.join(str(tmp_path), 'steps%d.txt' % r)).read()) for r in range(2)]
    assert steps[0] == steps[1] == 2 * 2, steps            # ceil(ceil(5/2)/2) = 2 batches per rank and epoch, 2 epochs


def _two_rank_worker(rank, world, out, port):
    import torch.distributed as dist
    from torch.utils.data import DataLoader, Dataset
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)

    class DS(Dataset):
        def __len__(self):
            return 5

        def __getitem__(self, i):
            return {'x': torch.tensor([float(i)])}

    class Stub(_Stub):
        def train_step(self, batch, reducer=None, lr=None):
            t = batch['x'].sum().reshape(1).clone()
            dist.all_reduce(t)                               # the per-step collective of the real model
            return super().train_step(batch, reducer, lr)

    opt = load_option()
    opt.epoch, opt.scheduler, opt.sync_batch = 2, 'none', False
    m = Stub()
    import dualpixelface_amd.distributed as dd
    dd.make_reducer = lambda model: None                     # the stub has no flat gradient arena
    tr = Trainer(opt, out, rank=rank, world_size=world)
    tr.fit(m, DataLoader(DS(), batch_size=2, shuffle=True), None)
    open(os.path.join(out, 'steps%d.txt' % rank), 'w').write(str(len(m.steps)))
    dist.destroy_process_group()
