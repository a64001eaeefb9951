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
    """The reference's metric hook (src/metric/metric_selector.py:7-39): one benchmark object per name in
    ``option.model.metric_type``; ``forward`` returns {name: metric row} and logs it, ``viewer`` prints the running means."""

    def __init__(self, option):
        from .metrics import BENCHMARKS
        self.metric_func, self.metric_name = [], []
        for name in list(getattr(option.model, 'metric_type', [])):
            if name not in BENCHMARKS:
                raise NotImplementedError('wrong metric type : %s' % name)
            self.metric_func.append(BENCHMARKS[name](option))
            self.metric_name.append(name)

    def forward(self, pred, batch, log=True, target_type='disp'):
        with torch.no_grad():
            return {name: func.measure(pred, batch, log, target_type) for name, func in zip(self.metric_name, self.metric_func)}

    def viewer(self):
        for name, func in zip(self.metric_name, self.metric_func):
            print('metric_type = %s' % name)
            results, table = func.get_value(use_chart=True)
            if table is not None:
                print(table)
