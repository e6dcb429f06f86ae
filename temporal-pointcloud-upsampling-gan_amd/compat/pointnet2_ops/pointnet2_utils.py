"""``pointnet2_ops.pointnet2_utils`` served by the MI355X HIP kernels.

Same names, argument order, dtypes and layouts as upstream so that the reference
imports (`discriminator.py:7-8`, `gcn_lib/pointnet/gcn.py:9`) resolve unchanged.
"""
import torch
import torch.nn as nn

import tpgan_amd.ops as _ops

furthest_point_sample = _ops.furthest_point_sample
gather_operation = _ops.gather_operation
ball_query = _ops.ball_query
grouping_operation = _ops.grouping_operation
three_nn = _ops.three_nn
three_interpolate = _ops.three_interpolate


class QueryAndGroup(nn.Module):
    """ball_query + group(xyz) - centre + group(features), concatenated on channels.

    forward(xyz (B,N,3), new_xyz (B,S,3), features (B,C,N)|None) -> (B,3+C,S,nsample).
    Built by the reference at discriminator.py:190."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz


class GroupAll(nn.Module):
    """forward(xyz, new_xyz(ignored), features) -> (B,3+C,1,N).  discriminator.py:192."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        return grouped_xyz
