"""pytorch_lightning.loggers.TensorBoardLogger as main.py:29 uses it.  Scalars go to TensorBoard event files when
torch.utils.tensorboard is importable, and always to <save_dir>/scalars.jsonl."""
import json
import os


class TensorBoardLogger(object):
    def __init__(self, save_dir, name='default', version=None, **ignored):
        self.save_dir = str(save_dir)
        self.name = name
        self.version = version

    @property
    def log_dir(self):
        return self.save_dir

    def log_history(self, history):
        os.makedirs(self.save_dir, exist_ok=True)
        with open(os.path.join(self.save_dir, 'scalars.jsonl'), 'a') as fh:
            for rec in history:
                fh.write(json.dumps(rec) + '\n')
        try:
            from torch.utils.tensorboard import SummaryWriter
        except Exception:
            return
        writer = SummaryWriter(self.save_dir)
        for rec in history:
            step = rec.get('step', rec.get('epoch', 0))
            for key, val in rec.items():
                if isinstance(val, (int, float)) and key not in ('step', 'epoch'):
                    writer.add_scalar(key, val, step)
        writer.close()
