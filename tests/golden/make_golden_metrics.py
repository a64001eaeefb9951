"""Golden vectors for the evaluation metrics: runs the REFERENCE's own absolute_dp / normal_dp metric functions
(/root/reference/src/metric/{absolute_dp,normal_dp}/metric.py -- numpy / torch only) on seeded inputs and stores inputs + outputs.
affine_dp needs TensorFlow (absent here) and is therefore not covered.  Run in the build container only:
    python tests/golden/make_golden_metrics.py
"""
import os
import runpy
import warnings

import numpy as np
import torch

REF = '/root/reference/src/metric'
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    warnings.simplefilter('ignore')
    g = torch.Generator().manual_seed(77)
    B, H, W = 2, 24, 36
    gt = torch.rand(B, H, W, generator=g) * 300 + 400
    pred = gt * (1 + 0.02 * torch.randn(B, H, W, generator=g))
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    absm = runpy.run_path(os.path.join(REF, 'absolute_dp', 'metric.py'))
    abs_out = np.asarray(absm['compute_errors_test_depth'](gt.numpy(), pred.numpy(), mask.numpy(), 1.01), dtype=np.float64)
    gn = torch.randn(B, 3, H, W, generator=g)
    pn = gn + 0.3 * torch.randn(B, 3, H, W, generator=g)
    nm = runpy.run_path(os.path.join(REF, 'normal_dp', 'metric.py'))
    n_mean = float(nm['calNormalAcc'](gn, pn, mask.unsqueeze(1)))
    n_rmse = float(nm['calNormalAccRMSE'](gn, pn, mask.unsqueeze(1)))
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), gt=gt.numpy(), pred=pred.numpy(), mask=mask.numpy(), abs_out=abs_out,
                        gn=gn.numpy(), pn=pn.numpy(), normal_out=np.asarray([n_mean, n_rmse]))
    print('absolute_dp', abs_out, 'normal_dp', n_mean, n_rmse)


if __name__ == '__main__':
    main()
