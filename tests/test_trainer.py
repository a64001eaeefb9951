"""Trainer logic that needs no GPU: the reference's schedules and the checkpoint round trip on a stand-in model."""
import os

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


def test_rank_strided_sharding():
    opt = load_option()
    batches = list(range(7))
    seen = [list(Trainer(opt, '.', rank=r, world_size=3)._shard(batches)) for r in range(3)]
    assert seen == [[0, 3, 6], [1, 4], [2, 5]]
