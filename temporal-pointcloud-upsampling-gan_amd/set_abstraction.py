"""PointNet++ set abstraction, FlowNet3D-style flow embedding and the four discriminators.

Host-side mirror of the reference's `discriminator.py` (knn :13-21, ball_query_wrapper
:24-40, index_points :43-60, build_shared_mlp :63-78, _PointnetSAModuleBase :83-153,
MSGSetConv :156-200, SSGSetConv :203-232, FlowEmbedding :235-283, FlowModule :286-322,
ActionTempoDis :325-402, ActionSpatialDis :405-470, FluidTempoDis :473-559,
FluidSpatialDis :562-629) with identical parameter / buffer names.  FPS, ball query,
grouping, gather and the neighbour searches run on the HIP kernels.
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import ops


def knn(k, xyz1, xyz2):
    dist, idx = ops.neighbour_search(xyz1, xyz2, k)
    return ops.attach_dist_grad(xyz1, xyz2, dist, idx), idx


def ball_query_wrapper(radius, sample, xyz1, xyz2):
    """<=sample in-radius hits, the rest padded with the same-slot kNN index
    (discriminator.py:24-40).

    Both lists are ordered by the same canonical key (dist, idx), and the in-radius hits
    are exactly the candidates with dist < r^2, i.e. a PREFIX of the kNN list; replacing
    the -1 tail by the kNN tail therefore reproduces the kNN list itself.  One search
    instead of two searches plus a masked copy; `radius` only documents intent."""
    del radius
    return ops.neighbour_search(xyz1, xyz2, sample)[1]


def index_points(points, idx):
    """points (B,N,C), idx (B,S[,K]) -> (B,S[,K],C)."""
    B = points.shape[0]
    batch = torch.arange(B, device=points.device).view([B] + [1] * (idx.dim() - 1))
    return points[batch, idx]


def build_shared_mlp(mlp_spec: List[int], bn: bool = True, sn: bool = True, act_fn=None):
    act_fn = nn.ReLU(True) if act_fn is None else act_fn
    layers = []
    for i in range(1, len(mlp_spec)):
        conv = nn.Conv2d(mlp_spec[i - 1], mlp_spec[i], kernel_size=1, bias=not bn)
        layers.append(spectral_norm(conv) if sn else conv)
        if bn:
            layers.append(nn.BatchNorm2d(mlp_spec[i]))
        layers.append(act_fn)
    return nn.Sequential(*layers)


class QueryAndGroup(nn.Module):
    """ball_query + group(xyz) - centre + group(features) -> (B,3+C,S,ns)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ops.ball_query(self.radius, self.nsample, xyz, new_xyz)
        g_xyz = ops.grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        g_xyz = g_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return g_xyz
        g_f = ops.grouping_operation(features.float().contiguous(), idx)
        return torch.cat([g_xyz, g_f], dim=1) if self.use_xyz else g_f


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        g_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return g_xyz
        g_f = features.unsqueeze(2)
        return torch.cat([g_xyz, g_f], dim=1) if self.use_xyz else g_f


def replace_dummy_centres(xyz, centres, rng=np.random):
    """discriminator.py:115-130: FPS hits on 999-dummies are swapped for random indices.

    Quirk kept from the reference: the candidate pool is every point index that is not one
    of the dummy *slot positions* inside `centres` (not point ids), so a replacement may
    itself be a dummy; survivors keep their order, replacements go to the tail."""
    hit = torch.abs(index_points(xyz, centres.long())[:, :, 0] - 999) < 1e-4
    if not bool(torch.any(hit)):                                    # host sync, as upstream
        return centres
    N = xyz.shape[1]
    for b in range(xyz.shape[0]):
        slots = hit[b].nonzero().view(-1)
        if slots.shape[0] == 0:
            continue
        pool_mask = torch.ones(N, dtype=torch.bool, device=xyz.device)
        pool_mask[slots[slots < N]] = False
        pool = pool_mask.nonzero().view(-1)
        pick = rng.choice(np.arange(pool.shape[0]), slots.shape[0], replace=False)
        pick = torch.as_tensor(pick, device=xyz.device)
        centres[b] = torch.cat((centres[b][~hit[b]], pool[pick].to(centres.dtype)))
    return centres


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.mask_dummy = False
        # set False by a caller that KNOWS the cloud carries no 999-dummies (skips one
        # host sync per call without changing any result)
        self.check_dummies = True

    def sample_centres(self, xyz):
        centres = ops.furthest_point_sample(xyz, self.npoint)
        if self.mask_dummy and self.check_dummies:
            centres = replace_dummy_centres(xyz, centres)
        return centres

    def forward(self, xyz, features):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3)|None, new_features (B,C',npoint)."""
        xyz = xyz.contiguous()
        if self.npoint is not None:
            centres = self.sample_centres(xyz)
            new_xyz = ops.gather_operation(xyz.transpose(1, 2).contiguous(), centres)
            new_xyz = new_xyz.transpose(1, 2).contiguous()
        else:
            new_xyz = None
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            g = mlp(grouper(xyz, new_xyz, features))               # (B,C',npoint,ns)
            outs.append(F.max_pool2d(g, kernel_size=[1, g.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class MSGSetConv(_PointnetSAModuleBase):
    def __init__(self, npoint, radii, nsamples, mlps, act_fn=None, mask_dummy=False, bn=True,
                 use_xyz=True, sn=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.mask_dummy = bool(mask_dummy)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(build_shared_mlp(spec, bn, sn, act_fn=act_fn))


class SSGSetConv(MSGSetConv):
    def __init__(self, mlp, npoint=None, mask_dummy=None, radius=None, nsample=None, bn=True,
                 use_xyz=True, sn=True, act_fn=None):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample],
                         mask_dummy=mask_dummy, bn=bn, use_xyz=use_xyz, sn=sn, act_fn=act_fn)


class FlowEmbedding(nn.Module):
    """Correlate frame-1 points with their 32 nearest frame-2 points (discriminator.py:235-283)."""

    NSAMPLE = 32

    def __init__(self, in_channel, mlp, pooling="max", corr_func="concat", sn=False):
        super().__init__()
        if corr_func != "concat":
            raise NotImplementedError("only corr_func='concat' is defined by the reference")
        self.pooling, self.corr_func = pooling, corr_func
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel * 2 + 3
        for out_channel in mlp:
            conv = nn.Conv2d(last, out_channel, 1, bias=False)
            self.mlp_convs.append(spectral_norm(conv) if sn else conv)
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def forward(self, pos1, pos2, feature1, feature2, radius):
        """pos (B,3,N), feature (B,C,N) -> pos1, (B,mlp[-1],N)."""
        B, _, N = pos1.shape
        idx = ball_query_wrapper(radius, self.NSAMPLE, pos1.transpose(1, 2), pos2.transpose(1, 2))
        idx = idx.to(torch.int32).contiguous()
        pos_diff = ops.grouping_operation(pos2.float().contiguous(), idx) - pos1.view(B, -1, N, 1)
        feat2 = ops.grouping_operation(feature2.float().contiguous(), idx)
        x = torch.cat([pos_diff, feat2, feature1.view(B, -1, N, 1).expand(-1, -1, -1, self.NSAMPLE)], dim=1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = F.leaky_relu(bn(conv(x)))
        return pos1, torch.max(x, -1)[0]


class FlowModule(nn.Module):
    def __init__(self, in_feat, hidden_feat, out_feat, sequence_length, sn=False):
        super().__init__()
        if sequence_length < 1:
            raise Exception("Flow module only accepts sequence with length greater than 1")
        self.flow_emb_layers = nn.ModuleList()
        self.depth = sequence_length - 1
        if self.depth == 1:
            hidden_feat = out_feat
        for depth in range(sequence_length - 1):
            if depth == 0:
                spec = (in_feat, [in_feat, hidden_feat // 2, hidden_feat])
            elif depth == sequence_length - 2:
                spec = (hidden_feat, [hidden_feat, out_feat, out_feat])
            else:
                spec = (hidden_feat, [hidden_feat, hidden_feat // 2, hidden_feat])
            self.flow_emb_layers.append(FlowEmbedding(spec[0], spec[1], sn=sn))

    def forward(self, feature_lst, pos_lst, cutoff):
        assert len(feature_lst) == self.depth + 1
        feats = list(feature_lst)
        for depth in range(self.depth):
            layer = self.flow_emb_layers[depth]
            feats = [layer(pos_lst[l].contiguous(), pos_lst[l + 1].contiguous(), feats[l].contiguous(),
                           feats[l + 1].contiguous(), cutoff)[1] for l in range(len(feats) - 1)]
        assert len(feats) == 1
        return feats[0]


def _head(dims, drops):
    """sn-Linear -> BN1d -> LeakyReLU [-> Dropout] ... -> sn-Linear(.,1), reference indices kept."""
    layers = []
    for i in range(len(dims) - 2):
        layers += [spectral_norm(nn.Linear(dims[i], dims[i + 1])), nn.BatchNorm1d(dims[i + 1]), nn.LeakyReLU()]
        if drops[i]:
            layers.append(nn.Dropout(drops[i]))
    layers.append(spectral_norm(nn.Linear(dims[-2], dims[-1])))
    return nn.Sequential(*layers)


class _TempoDis(nn.Module):
    """Per-frame SA x2 -> FlowModule over the T frames -> GroupAll SA -> FC head."""

    flow_radius_scale = 1.0

    def _levels(self, pos_lst, feat_lst):
        feats, poss = [], []
        for i, pos in enumerate(pos_lst):
            f0 = (feat_lst[i] if feat_lst is not None else pos).transpose(1, 2).contiguous()
            p1, f1 = self.coarse_graining_module[0](pos, f0)
            poss.append(p1)
            feats.append(f1)
        feats2, poss2 = [], []
        for f, p in zip(feats, poss):
            p2, f2 = self.coarse_graining_module[1](p, f)
            feats2.append(f2)
            poss2.append(p2.permute(0, 2, 1))                      # (B,3,N) for FlowEmbedding
        return feats2, poss2

    def _forward(self, pos_lst, cutoff, feat_lst, width):
        if feat_lst is not None:
            assert len(feat_lst) == len(pos_lst)
        feats, poss = self._levels(pos_lst, feat_lst)
        f = self.flow_module(feats, poss, self.flow_radius_scale * cutoff)
        _, f = self.SA_pooling(poss[0].permute(0, 2, 1), f)
        return self.fc_layers(f.view(-1, width))


class ActionTempoDis(_TempoDis):
    def __init__(self, sequence_length, sn=True):
        super().__init__()
        self.coarse_graining_module = nn.ModuleList([
            SSGSetConv(npoint=512, radius=0.8, nsample=64, mlp=[3, 64, 64, 128], use_xyz=True, sn=sn),
            SSGSetConv(npoint=256, radius=1.2, nsample=32, mlp=[128, 128, 256], use_xyz=True, sn=sn)])
        self.flow_module = FlowModule(256, 256, 256, sequence_length, sn=sn)
        self.SA_pooling = SSGSetConv(mlp=[256, 256, 512], use_xyz=True, sn=sn)
        self.fc_layers = _head([512, 256, 64, 1], [0.3, 0.1])

    def forward(self, pos_lst, cutoff):
        return self._forward(pos_lst, cutoff, None, 512)


class FluidTempoDis(_TempoDis):
    flow_radius_scale = 20.0                                       # discriminator.py:552

    def __init__(self, sequence_length, sn=True):
        super().__init__()
        self.coarse_graining_module = nn.ModuleList([
            SSGSetConv(npoint=1024, radius=0.10, nsample=32, mlp=[3, 64, 128], use_xyz=True, sn=sn,
                       mask_dummy=True, act_fn=nn.LeakyReLU()),
            SSGSetConv(npoint=256, radius=0.20, nsample=32, mlp=[128, 128, 256], use_xyz=True, sn=sn,
                       act_fn=nn.LeakyReLU())])
        self.flow_module = FlowModule(256, 256, 256, sequence_length, sn=sn)
        self.SA_pooling = SSGSetConv(mlp=[256, 256, 256], use_xyz=True, sn=sn, act_fn=nn.LeakyReLU())
        self.fc_layers = _head([256, 256, 64, 1], [0.2, 0.0])

    def forward(self, pos_lst, cutoff, feat_lst=None):
        return self._forward(pos_lst, cutoff, feat_lst, 256)


class _SpatialDis(nn.Module):
    def _forward(self, pos, width):
        feature = None
        for sa in self.coarse_graining_module:
            pos, feature = sa(pos, pos.transpose(1, 2).contiguous() if feature is None else feature)
        _, feature = self.SA_pooling(pos, feature)
        return self.fc_layers(feature.view(-1, width))


class ActionSpatialDis(_SpatialDis):
    def __init__(self, sn=True):
        super().__init__()
        self.coarse_graining_module = nn.ModuleList([
            SSGSetConv(npoint=512, radius=0.3, nsample=32, mlp=[3, 64, 64, 128], use_xyz=True, sn=sn),
            SSGSetConv(npoint=256, radius=0.6, nsample=32, mlp=[128, 128, 128], use_xyz=True, sn=sn),
            SSGSetConv(npoint=128, radius=1.0, nsample=32, mlp=[128, 128, 256], use_xyz=True, sn=sn)])
        self.SA_pooling = SSGSetConv(mlp=[256, 256, 512], use_xyz=True, sn=sn)
        self.fc_layers = _head([512, 256, 64, 1], [0.3, 0.1])

    def forward(self, pos):
        return self._forward(pos, 512)


class FluidSpatialDis(_SpatialDis):
    def __init__(self, sn=True):
        super().__init__()
        lrelu = nn.LeakyReLU
        self.coarse_graining_module = nn.ModuleList([
            SSGSetConv(npoint=1024, radius=0.15, nsample=32, mlp=[3, 64, 128], use_xyz=True, sn=True,
                       mask_dummy=True, act_fn=lrelu()),
            SSGSetConv(npoint=512, radius=0.30, nsample=32, mlp=[128, 128, 128], use_xyz=True, sn=True,
                       act_fn=lrelu()),
            SSGSetConv(npoint=128, radius=0.60, nsample=16, mlp=[128, 128, 256], use_xyz=True, sn=True,
                       act_fn=lrelu())])
        self.SA_pooling = SSGSetConv(mlp=[256, 256, 256], use_xyz=True, sn=sn)
        self.fc_layers = _head([256, 256, 64, 1], [0.2, 0.0])

    def forward(self, pos):
        return self._forward(pos, 256)
