"""Optimiser / scheduler / metric selection with the reference's names (src/model/model_selector.py:31-58,
src/metric/metric_selector.py)."""
import torch


def optimizer_selector(params, option):
    if option.optim == 'adam':
        return torch.optim.Adam(params, lr=float(option.init_lr), betas=(0.9, 0.999), eps=1e-5)
    if option.optim == 'sgd':
        return torch.optim.SGD(params, lr=float(option.init_lr), momentum=0.9, weight_decay=2e-4)
    if option.optim == 'rmsprop':
        return torch.optim.RMSprop(params, lr=float(option.init_lr), eps=1e-5)
    raise NotImplementedError('optimizer is not defined, please check your optimizer configuration !')


def scheduler_selector(optimizer, option):
    if option.scheduler == 'steplr':
        return torch.optim.lr_scheduler.StepLR(optimizer, 35, 0.5)
    if option.scheduler == 'explr':
        return torch.optim.lr_scheduler.ExponentialLR(optimizer, 0.5)
    if option.scheduler == 'cosanneal':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, 500, 1e-6)
    if option.scheduler == 'none':
        return None
    raise NotImplementedError('scheduler is not defined, please check your scheduler configuration !')


class metric_selector(object):
    """Hook object only: the evaluation metrics (absolute_dp / affine_dp / normal_dp, CPU + TensorFlow in the
    reference) are outside the training hot path."""

    def __init__(self, option):
        self.names = list(getattr(option.model, 'metric_type', []))

    def forward(self, results, batch):
        return {}

    def viewer(self):
        return None
