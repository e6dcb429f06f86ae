"""``chamferdist.ChamferDistance`` served by the MI355X HIP kernels.

Reference call sites: loss.py:125-127,176-181 (`cd(a, b, bidirectional=True)`).
Semantics of chamferdist 1.0: per-point squared nn distance, summed over points,
then `reduction` over the batch ("mean" default | "sum" | None)."""
import torch
import torch.nn as nn

import tpgan_amd.ops as _ops


class ChamferDistance(nn.Module):
    def forward(self, source_cloud, target_cloud, bidirectional=False, reverse=False,
                reduction="mean"):
        if not isinstance(source_cloud, torch.Tensor) or not isinstance(target_cloud, torch.Tensor):
            raise TypeError("Expected input type torch.Tensor")
        if source_cloud.device != target_cloud.device:
            raise ValueError("Source and target clouds must be on the same device")
        if source_cloud.dim() != 3 or target_cloud.dim() != 3:
            raise ValueError("clouds must be (B, N, 3)")
        if source_cloud.shape[0] != target_cloud.shape[0]:
            raise ValueError("Source and target pointclouds must have the same batchsize")
        if source_cloud.shape[2] != target_cloud.shape[2]:
            raise ValueError("Source and target pointclouds must have the same dimensionality")
        if bidirectional and reverse:
            import warnings
            warnings.warn("Both bidirectional and reverse set to True. bidirectional takes precedence")
        if reduction not in ("sum", "mean", None):
            raise ValueError('Reduction must either be "sum" or "mean" or None')
        d1, d2, _, _ = _ops.chamfer_nn(source_cloud, target_cloud)
        fwd = d1.sum(1)  # (B,)
        bwd = d2.sum(1)
        if reduction == "sum":
            fwd, bwd = fwd.sum(), bwd.sum()
        elif reduction == "mean":
            fwd, bwd = fwd.mean(), bwd.mean()
        if bidirectional:
            return fwd + bwd
        if reverse:
            return bwd
        return fwd
