"""PointNet-layout (B,C,N,k) graph convolutions of the generator.

Host-side mirror of the reference's `gcn_lib/pointnet/gcn.py` (knn_query :13-22,
Dilated/DilatedKnnGraph :48-93, build_shared_mlp :96-120, conv_bn_layer :123-147,
EdgeConv :150-212, IDGCNLayer :215-279).  Module/attribute names and Sequential
indices are kept identical so state dicts interchange with the reference
(SURVEY.md Appendix B); the neighbour search and grouping go to the HIP kernels.
"""
from typing import List

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops

_NORMS = ("batch", "ins", "none")


def knn_query(k, xyz1, xyz2=None):
    """(dist, idx int64) of the k nearest xyz2 rows per xyz1 row (gcn.py:13-22)."""
    if xyz2 is None:
        xyz2 = xyz1
    dist, idx = ops.neighbour_search(xyz1, xyz2, k)
    return ops.attach_dist_grad(xyz1, xyz2, dist, idx), idx


def radius_query(radius, sample, xyz1, xyz2=None, knn_padding=True):
    """gcn.py:25-45 (defined there, never called): <=sample hits within radius, -1 slots
    optionally replaced by the same-slot kNN index."""
    if xyz2 is None:
        xyz2 = xyz1
    dist, idx = ops.neighbour_search(xyz1, xyz2, sample, r=radius)
    if knn_padding:
        _, kidx = ops.neighbour_search(xyz1, xyz2, sample)
        idx = torch.where(idx == -1, kidx, idx)
    return dist, idx


def _norm_name(bn, insn):
    if insn and bn:
        raise Exception("Cant use batch normalization and instance normalization at the same time")
    return "batch" if bn else ("ins" if insn else "none")


def _conv1x1(cin, cout, norm, sn):
    # The reference's bias flag is inverted (gcn.py:98,102-106,125,128-132): a conv followed
    # by batch/instance norm HAS a bias, a bare conv (norm == 'none') has NONE.
    if norm not in _NORMS:
        raise Exception(f"Unsupported normalization: {norm}")
    conv = nn.Conv2d(cin, cout, kernel_size=1, bias=norm in ("batch", "ins"))
    layers = [spectral_norm(conv) if sn else conv]
    if norm == "batch":
        layers.append(nn.BatchNorm2d(cout))
    elif norm == "ins":
        layers.append(nn.InstanceNorm2d(cout))
    return layers


def build_shared_mlp(mlp_spec: List[int], norm: str = "batch", sn: bool = False):
    layers = []
    for i in range(1, len(mlp_spec)):
        layers += _conv1x1(mlp_spec[i - 1], mlp_spec[i], norm, sn)
        layers.append(nn.LeakyReLU(0.2))
    return nn.Sequential(*layers)


def conv_bn_layer(in_feat, out_feat, act=False, norm="batch", sn=False):
    layers = _conv1x1(in_feat, out_feat, norm, sn)
    if act:
        layers.append(nn.LeakyReLU(0.2))
    return nn.Sequential(*layers)


class Dilated(nn.Module):
    """Every `dilation`-th entry of a k-NN list (gcn.py:48-72)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation, self.stochastic, self.epsilon = k, dilation, stochastic, epsilon

    def forward(self, edge_index):
        if self.stochastic and self.training and torch.rand(1) < self.epsilon:
            num = self.k * self.dilation
            randnum = torch.randperm(num)[:self.k]
            return edge_index[:, :, randnum]
        return edge_index[:, :, ::self.dilation]


class DilatedKnnGraph(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation = k, dilation
        self._dilated = Dilated(k, dilation, stochastic, epsilon)

    def forward(self, x):
        """x (B,N,C) -> (B,N,k//dilation) int64."""
        _, idx = ops.neighbour_search(x, x, self.k)
        return self._dilated(idx)


_AGGREGATORS = {
    "sum": lambda y: torch.sum(y, dim=-1, keepdim=True),
    "max": lambda y: torch.max(y, dim=-1, keepdim=True)[0],
    "min": lambda y: torch.min(y, dim=-1, keepdim=True)[0],
    "mean": lambda y: torch.mean(y, dim=-1, keepdim=True),
}


class EdgeConv(nn.Module):
    """kNN -> group -> node/edge affine -> shared MLP -> aggregate over k (gcn.py:150-212)."""

    def __init__(self, in_feat, out_feat, k=9, dilation=1, mlp_layer=True, aggregate="max",
                 bn=True, insn=False, sn=False, **kwargs):
        super().__init__()
        self.norm = _norm_name(bn, insn)
        self.k = k // dilation
        self.dilated_knn_graph = DilatedKnnGraph(k, dilation, **kwargs)
        half = out_feat // 2
        self.edge_affine = conv_bn_layer(in_feat, half, act=True, norm=self.norm, sn=sn)
        self.node_affine = conv_bn_layer(in_feat, half, act=True, norm=self.norm, sn=sn)
        if mlp_layer:
            self.mlp = build_shared_mlp([half, half, out_feat], norm=self.norm, sn=sn)
        else:
            self.mlp = conv_bn_layer(half, out_feat, norm=self.norm, sn=sn, act=False)
        if aggregate not in _AGGREGATORS:
            raise Exception(f"Unsupported aggregation mode {aggregate}")
        self.aggregate_fn = _AGGREGATORS[aggregate]

    def forward(self, feat, pos=None):
        if feat.dim() == 4 and feat.shape[-1] == 1:
            feat = feat.squeeze(-1)
        feat = feat.contiguous()                                   # (B,C,N)
        search_in = pos if pos is not None else feat.transpose(1, 2)
        knn_idx = self.dilated_knn_graph(search_in).to(torch.int32).contiguous()
        grouped = ops.grouping_operation(feat.float(), knn_idx)   # (B,C,N,k)
        edge = grouped - feat.unsqueeze(-1)
        out = self.node_affine(grouped) + self.edge_affine(edge)
        return self.aggregate_fn(self.mlp(out))                    # (B,C_out,N,1)


class IDGCNLayer(nn.Module):
    """Inception-DenseGCN block (gcn.py:215-279)."""

    def __init__(self, in_feats, out_feats, bn=True, insn=False, ln=False, sn=False, residual=True):
        super().__init__()
        self.norm = _norm_name(bn, insn)
        q = in_feats // 4
        self.btn = conv_bn_layer(in_feats, q, act=False, norm=self.norm, sn=sn)
        self.GCN1 = EdgeConv(q, q, k=20, dilation=1, aggregate="max", mlp_layer=True, bn=bn, insn=insn, sn=sn)
        self.GCN2 = EdgeConv(q, q, k=20, dilation=2, aggregate="max", mlp_layer=True, bn=bn, insn=insn, sn=sn)
        self.decoder = conv_bn_layer(q * 3, out_feats, act=True, norm=self.norm, sn=sn)
        self.use_layernorm = ln
        if ln:
            self.layernorm = nn.LayerNorm([out_feats])
        self.residual = residual
        if residual:
            self.skip_layer = conv_bn_layer(in_feats, out_feats, act=False, norm=self.norm, sn=sn)

    def forward(self, feature):                                    # (B,C,N,1)
        skip = self.skip_layer(feature) if self.residual else None
        low = self.btn(feature).squeeze(-1).contiguous()           # (B,C/4,N)
        _, idx = ops.neighbour_search(low.transpose(1, 2), low.transpose(1, 2), 9)
        local = ops.grouping_operation(low.float(), idx.to(torch.int32).contiguous())
        local_max = torch.max(local, dim=-1, keepdim=True)[0]
        f1 = self.GCN1(low)
        f2 = self.GCN2(low)
        out = self.decoder(torch.cat([local_max, f1, f2], dim=1))
        if self.use_layernorm:
            B, C, N, _ = out.shape
            out = self.layernorm(out.squeeze(-1).permute(0, 2, 1).reshape(-1, C))
            out = out.reshape(B, N, C).permute(0, 2, 1).unsqueeze(-1).contiguous()
        if self.residual:
            out = out + skip
        return out
