"""pytorch_lightning.callbacks names used by main.py:9-10,32-41.  The native trainer writes one checkpoint per epoch into the
ModelCheckpoint's `dirpath` (PL's `checkpoint_epoch=NN.ckpt` naming, `save_top_k=-1, period=1` semantics); the learning rate is part
of every log record, so LearningRateMonitor has nothing left to do."""


class Callback(object):
    pass


class ModelCheckpoint(Callback):
    def __init__(self, dirpath=None, filename=None, save_top_k=-1, period=1, every_n_epochs=None, monitor=None, **ignored):
        self.dirpath = str(dirpath) if dirpath is not None else None
        self.filename = filename
        self.save_top_k = save_top_k
        self.period = every_n_epochs if every_n_epochs is not None else period


class LearningRateMonitor(Callback):
    def __init__(self, logging_interval=None, log_momentum=False, **ignored):
        self.logging_interval = logging_interval
