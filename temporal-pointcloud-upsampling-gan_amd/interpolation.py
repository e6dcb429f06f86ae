"""`gcn_lib.interpolation` of the reference, on the fused HIP kernel (csrc/cubic_interp.hip).

Only what the training path uses is provided: `cubic_interpolation` (train_step_final.py:61) and
the kernels it is built from.  The reference needs DGL for this file; this one does not."""
import numpy as np
import torch

from . import ops

DT = 0.025      # train_step_final.py:7


def bicubic_kernel(r, cutoff):
    """gcn_lib/interpolation.py:94-104."""
    coeff = 8. / (np.pi * cutoff ** 3)
    q = r / cutoff
    ker = torch.zeros_like(r)
    m1 = (q >= 0) & (q <= 0.5)
    m2 = (q > 0.5) & (q <= 1)
    ker = torch.where(m1, 6. * (q ** 3 - q ** 2) + 1., ker)
    ker = torch.where(m2, 2. * (1. - q) ** 3, ker)
    return ker * coeff


def cubic_interpolation(query_pos, field, pos, cutoff):
    """Same signature as the reference's (2-D per-sample tensors) and also batched 3-D."""
    return ops.cubic_interpolation(query_pos, field, pos, cutoff)


def interpolate_vel_lst(pred_pos_lst, gt_pos_lst, gt_vel_lst, opt, furthest_distance):
    """train_step_final.py:51-66: advection features of the real frames (vel * DT) and their
    interpolation at the predicted positions -- all frames and samples in ONE launch."""
    T, B = len(pred_pos_lst), pred_pos_lst[0].shape[0]
    with torch.no_grad():
        gt_adv = [v * DT for v in gt_vel_lst]
        pred_adv = ops.cubic_interpolation(torch.cat(list(pred_pos_lst), 0), torch.cat(gt_adv, 0),
                                           torch.cat(list(gt_pos_lst), 0), 1.6 * opt.R / furthest_distance)
    return gt_adv, list(pred_adv.view(T, B, *pred_adv.shape[1:]).unbind(0))
