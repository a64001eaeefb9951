"""Loss hooks with the reference's contract (src/loss/loss_selector.py:7-42) over the fused HIP loss kernel.

``loss_selector(option).forward(results, batch)`` -> ``{'<name>_loss': ..., 'abvalue': ..., 'final_loss': ...}``.
Class names follow the reference's lookup rule ``<NAME.upper()>Loss`` (loss_selector.py:25).
"""
import torch

from . import ops


class SMOOTHL1Loss(object):
    """Masked smooth-L1 over the disparity heads (src/loss/depth/smoothL1.py:15-49, 'given' conversion)."""

    def __init__(self, option):
        if option.dataset.dp_conversion != 'given':
            # (the reference's own 'least_square' branch cannot run on a batch with a mask -- which every FaceDP batch has: its target is
            # [B, 1, H, W] (smoothL1.py:29-30) and is then indexed with the [B, H, W] mask (smoothL1.py:38): IndexError)
            raise NotImplementedError("dp_conversion='least_square' (scipy lsq_linear on the host; broken in the reference for masked batches) is not built")
        self.weights = list(option.model.loss_weight)

    def head_weights(self, n):
        return [1.0] if n == 1 else self.weights[:n]

    def forward(self, preds, batch, target_type='disp'):
        if target_type != 'disp':
            raise NotImplementedError('only the disparity target is on the hot path')
        pd, gt = preds['pred_depth'], batch['disp']
        if batch.get('conf') is not None:
            # confidence-weighted variant (smoothL1.py:33-36): both sides of the difference are multiplied by the per-pixel confidence.  No
            # loader of the reference emits 'conf', so this is two elementwise products in front of the fused loss kernel, not a kernel of its own
            pd, gt = pd * batch['conf'].unsqueeze(1), gt * batch['conf']
        mask = batch['mask'] if 'mask' in batch else torch.ones_like(batch['disp'])
        out = ops.stereo_losses(pd, None, gt, None, mask, self.head_weights(pd.shape[1]), 1.0, 0.0)
        return {'loss': out[0], 'abvalue': batch['abvalue']}


class COSINELoss(object):
    """Masked per-channel "cosine" loss (src/loss/normal/cosine.py:35-53)."""

    def __init__(self, option):
        self.weights = list(option.model.loss_weight)

    def forward(self, preds, batch, target_type=None):
        pn = preds['pred_normal']
        if pn.shape[1] != 1:
            raise NotImplementedError('one normal prediction per sample (mainmodel.py:95)')
        mask = batch['mask'] if 'mask' in batch else torch.ones_like(batch['normal'][:, 0])
        out = ops.stereo_losses(None, pn[:, 0], None, batch['normal'], mask, [], 0.0, 1.0)
        return {'loss': out[1]}


_BANK = {'smoothL1': SMOOTHL1Loss, 'cosine': COSINELoss}


class loss_selector(object):
    def __init__(self, option):
        assert len(option.model.loss_type) == len(option.model.lambdas)
        self.loss_func, self.loss_name, self.lambda_ = [], [], []
        for name, lam in zip(option.model.loss_type, option.model.lambdas):
            if name not in _BANK:
                raise NotImplementedError('wrong loss type : %s' % name)
            self.loss_func.append(_BANK[name](option))
            self.loss_name.append(name)
            self.lambda_.append(lam)

    def forward(self, results, batch, target_type='disp'):
        # shipped configuration (stereodpnet/config.json:2,4): one fused reduction pass for both losses
        if self.loss_name == ['smoothL1', 'cosine'] and results.get('pred_normal') is not None and 'mask' in batch and batch.get('conf') is None:
            pd, pn = results['pred_depth'], results['pred_normal']
            out = ops.stereo_losses(pd, pn[:, 0], batch['disp'], batch['normal'], batch['mask'],
                                    self.loss_func[0].head_weights(pd.shape[1]), self.lambda_[0], self.lambda_[1])
            return {'smoothL1_loss': out[0], 'abvalue': batch['abvalue'], 'cosine_loss': out[1], 'final_loss': out[2]}
        result, total = {}, []
        for name, lam, fn in zip(self.loss_name, self.lambda_, self.loss_func):
            out = fn.forward(results, batch, target_type)
            result[name + '_loss'] = out['loss']
            if 'abvalue' in out:
                result['abvalue'] = out['abvalue']
            total.append(lam * out['loss'])
        result['final_loss'] = sum(total)
        return result
