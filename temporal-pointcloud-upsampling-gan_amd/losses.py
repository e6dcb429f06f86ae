"""Position losses of the train step: Chamfer + mask supervision.

Host-side mirror of the reference's `loss.py` (index_points :10-27, chamfer_distance_loss
:121-128, tpugan_sr_loss :168-183, masking_loss :253-275).  The remaining functions of
that file (EMD, repulsion, density ...) are not called by the train step and are out of
scope.
"""
import torch

from . import ops


def chamfer_distance(a, b, bidirectional=True, reduction="mean"):
    """Sum over points of squared nn distance, `reduction` over the batch (chamferdist 1.0)."""
    d1, d2, _, _ = ops.chamfer_nn(a, b)
    fwd, bwd = d1.sum(1), d2.sum(1)
    if reduction == "mean":
        fwd, bwd = fwd.mean(), bwd.mean()
    elif reduction == "sum":
        fwd, bwd = fwd.sum(), bwd.sum()
    return fwd + bwd if bidirectional else fwd


def chamfer_distance_loss(pcd1_pos, pcd2_pos):
    """Unbatched (N,3) clouds (loss.py:121-128)."""
    return chamfer_distance(pcd1_pos[None], pcd2_pos[None])


def masking_loss(pos_gt, pos_input, binary_mask, particle_radius):
    """L1 between the predicted mask and "my nearest gt point (within 1.9 r) has more than 3
    gt neighbours within 1.4 r" (loss.py:253-275)."""
    _, nn_idx = ops.neighbour_search(pos_input, pos_gt, 1, r=particle_radius * 1.9)
    _, self_idx = ops.neighbour_search(pos_gt, pos_gt, 16, r=particle_radius * 1.4)
    crowded = ((self_idx != -1).sum(dim=-1) > 3).to(binary_mask.dtype)      # (B,Ngt)
    # inputs without any gt neighbour carry idx -1 -> they read the appended zero column
    crowded = torch.cat([crowded, crowded.new_zeros(crowded.shape[0], 1)], dim=1)
    target = torch.gather(crowded, 1, nn_idx.squeeze(-1) % crowded.shape[1]).unsqueeze(-1)
    return torch.nn.functional.l1_loss(binary_mask, target)


def tpugan_sr_loss(w1, gt_pcd_pos, pred_pcd_pos, input_pcd_pos, mask, particle_radius, n_iter):
    """Chamfer(gt, pred) + w1 * masking loss; the mask term is the constant 1.0 for the first
    10 iterations or when w1 == 0 (loss.py:168-183)."""
    if n_iter > 10 and w1 != 0:
        m_loss = masking_loss(gt_pcd_pos, input_pcd_pos, mask, particle_radius)
    else:
        m_loss = torch.ones(1, device=pred_pcd_pos.device)
    if gt_pcd_pos.dim() == 2:
        gt_pcd_pos = gt_pcd_pos.unsqueeze(0)
    if pred_pcd_pos.dim() == 2:
        pred_pcd_pos = pred_pcd_pos.unsqueeze(0)
    cd = chamfer_distance(gt_pcd_pos, pred_pcd_pos)
    return cd + w1 * m_loss, cd, m_loss
