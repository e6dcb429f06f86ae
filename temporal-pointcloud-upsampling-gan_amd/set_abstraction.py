"""PointNet++ set abstraction, FlowNet3D-style flow embedding and the four discriminators.

Host-side mirror of the reference's `discriminator.py` (knn :13-21, ball_query_wrapper
:24-40, index_points :43-60, build_shared_mlp :63-78, _PointnetSAModuleBase :83-153,
MSGSetConv :156-200, SSGSetConv :203-232, FlowEmbedding :235-283, FlowModule :286-322,
ActionTempoDis :325-402, ActionSpatialDis :405-470, FluidTempoDis :473-559,
FluidSpatialDis :562-629) with identical parameter / buffer names.  FPS, ball query and the
neighbour searches run on the HIP kernels.

Inner loop, MI355X-first (same results up to fp32 rounding, pinned by tests/golden):
the reference builds cat([group(xyz) - centre, group(feat)]) of shape (B,3+C,S,ns) and runs
the first spectral-norm conv on all S*ns grouped positions.  Here the first conv (same
weight, same single power iteration per call) is applied to the N un-grouped rows
[xyz | feat] and to the S centres, and the rows of its output are gathered:

    y1[b,s,j,:] = U[b, idx[b,s,j], :] - Q[b,s,:]      U = W1 [xyz|feat],  Q = W1[:, :3] centre

(ops.row_combine, one coalesced row-copy kernel; ns-fold fewer first-layer FLOPs, no
(B,3+C,S,ns) tensor).  The remaining layers are GEMMs on channels-last rows with
BatchNorm over the (B*S*ns) row axis -- the same statistics as BatchNorm2d over (B,S,ns).
"""
import contextlib
import os
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import ops
from .graph_conv import amp_dtype, no_autocast, reference_order, rows_first, rows_matmul, rows_matmul_seg  # noqa: F401


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


# weights prepared for the current discriminator forward: id(module) -> list of 2-D weights, one
# per upcoming call (filled by sn_prefetch, consumed by sn_weight)
_SN_READY = {}


# weights computed AHEAD of a forward (sn_prepare): id(module) -> queue of per-forward lists
_SN_PREPARED = {}


def _sn_compute(todo, training):
    with torch.autocast(device_type=todo[0][0].weight_orig.device.type, enabled=False):
        ws = ops.spectral_normalize_many([m for m, _ in todo], [k for _, k in todo], training)
    return [[w.view(m.weight_orig.shape) for w in lst] for (m, _), lst in zip(todo, ws)]


def sn_prepare(modules_and_uses, training):
    """Compute NOW, on the current stream, the weights a later forward's `sn_prefetch` would compute
    (same arguments): the power iterations depend on the weights alone, so a caller whose stream is
    about to wait for something else (an index plan) can run them in that gap instead of at the
    head of the forward (170 us of one-workgroup-per-module work).  Forwards consume prepared sets
    in the order they were prepared; the caller keeps that the order of the module calls and calls
    `sn_discard_prepared` if a prepared forward does not happen."""
    todo = [(m, k) for m, k in modules_and_uses if hasattr(m, "weight_orig") and k > 0]
    if todo and rows_first():
        for (m, _), lst in zip(todo, _sn_compute(todo, training)):
            _SN_PREPARED.setdefault(id(m), []).append(lst)


def sn_discard_prepared():
    _SN_PREPARED.clear()


@contextlib.contextmanager
def sn_prefetch(modules_and_uses, training):
    """Compute, in ONE kernel launch, every spectrally-normalised weight the enclosed forward
    will ask for: modules_and_uses = [(module, number_of_calls)], in any order.  Weights prepared
    ahead for this forward (sn_prepare) are taken instead."""
    todo = [(m, k) for m, k in modules_and_uses if hasattr(m, "weight_orig") and k > 0]
    if todo and rows_first():
        ahead = [m for m, _ in todo if _SN_PREPARED.get(id(m))]
        if ahead:
            assert len(ahead) == len(todo), "spectral-norm weights prepared for some modules of a forward only"
            for m, k in todo:
                lst = _SN_PREPARED[id(m)].pop(0)
                assert len(lst) == k, "prepared spectral-norm weights do not match the forward's calls"
                _SN_READY[id(m)] = lst
        else:
            for (m, _), lst in zip(todo, _sn_compute(todo, training)):
                _SN_READY[id(m)] = lst
    try:
        yield
    finally:
        for m, _ in todo:
            _SN_READY.pop(id(m), None)


def sn_weight(module):
    """The weight a spectrally-normalised conv / linear would use in this forward: one power
    iteration (training mode) + W / sigma, computed by ONE fused kernel on the module's own
    `weight_orig` / `weight_u` / `weight_v` (ops.spectral_normalize) instead of PyTorch's
    forward pre-hook (~12 launches forward, as many backward).  Always fp32, never autocast.
    Modules without spectral norm return their plain weight."""
    if not hasattr(module, "weight_orig"):
        return module.weight
    ready = _SN_READY.get(id(module))
    if ready:
        return ready.pop(0)
    with torch.autocast(device_type=module.weight_orig.device.type, enabled=False):
        return ops.spectral_normalize(module.weight_orig, module.weight_u, module.weight_v, module.training)


def conv_weight2d(conv):
    """(Cout,Cin) weight of a 1x1 conv as its module call would see it (exactly one power
    iteration per call, like the reference's one module call)."""
    w = sn_weight(conv)
    return w.view(w.shape[0], -1)


def bn_rows(bn, x):
    """BatchNorm2d/1d module applied to rows (P,C): statistics over the P rows, i.e. over
    (B,S,ns) of the reference's (B,C,S,ns) tensor; running stats and num_batches_tracked
    updated exactly like the module's own forward."""
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    factor = 0.0 if bn.momentum is None else bn.momentum
    if bn.momentum is None and bn.training and bn.track_running_stats:
        factor = 1.0 / float(bn.num_batches_tracked)
    use_batch_stats = bn.training or (bn.running_mean is None and bn.running_var is None)
    y = F.batch_norm(x, bn.running_mean if not bn.training or bn.track_running_stats else None,
                     bn.running_var if not bn.training or bn.track_running_stats else None,
                     bn.weight, bn.bias, use_batch_stats, factor, bn.eps)
    return y.view(shape)


def _act_slope(m):
    """LeakyReLU slope of an activation module (ReLU = 0), or None if it is something else."""
    if isinstance(m, nn.ReLU):
        return 0.0
    if isinstance(m, nn.LeakyReLU):
        return float(m.negative_slope)
    return None


def _fusable(x, K):
    C = x.shape[-1]
    return x.dtype in (torch.float32, torch.bfloat16) and C % 8 == 0 and C <= 1024 and K <= 256


def bn_act_rows(bn, x, slope, K=0, nseg=1, mean_shift=None):
    """Fused BatchNorm + LeakyReLU(slope) [+ max over each group of K rows] on rows (P,C)
    (ops.row_bn_act, csrc/rowbn.hip); module state handled like nn.BatchNorm's own forward.
    nseg > 1: the rows are nseg equal blocks = nseg successive calls of `bn` in one launch.
    mean_shift: bias of the preceding conv that was left out of x (ops.row_bn_act)."""
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    momentum = 0.0 if bn.momentum is None else bn.momentum
    nbt = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn.momentum is None:                    # cumulative average: the factor is needed on the host
            assert nseg == 1, "segmented BatchNorm needs a fixed momentum"
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
        else:
            nbt = bn.num_batches_tracked           # incremented by the statistics kernel itself
    track = bn.track_running_stats and bn.running_mean is not None
    return ops.row_bn_act(x, bn.weight, bn.bias, bn.running_mean if track else None,
                          bn.running_var if track else None, training, momentum, bn.eps, slope, K,
                          out_dtype=amp_dtype(x), num_batches_tracked=nbt, nseg=nseg, mean_shift=mean_shift)


def conv_weights_seg(conv, nseg):
    """(nseg, Cout, Cin): the weights of `nseg` successive calls of a 1x1 conv (one power
    iteration each when it is spectrally normalised)."""
    if not hasattr(conv, "weight_orig"):
        w = conv.weight
        return w.view(1, w.shape[0], -1).expand(nseg, -1, -1)
    return torch.stack([conv_weight2d(conv) for _ in range(nseg)])


def _segmentable(layers, x, K):
    """Can `nseg` calls of this tail run as ONE segmented pass?  Every conv must be followed by a
    BatchNorm + (Leaky)ReLU pair the fused kernel takes (training-mode statistics, fixed
    momentum), and the rows must be on the GPU."""
    if not (x.is_cuda and rows_first()):
        return False
    n = len(layers)
    for i, m in enumerate(layers):
        if isinstance(m, nn.Conv2d):
            if not (i + 2 < n and isinstance(layers[i + 1], nn.BatchNorm2d)):
                return False
            if m.out_channels % 8 or m.out_channels > 1024:
                return False
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            if not (m.training and m.momentum is not None and i + 1 < n and _act_slope(layers[i + 1]) is not None):
                return False
        elif _act_slope(m) is None:
            return False
    return K <= 256 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1024


def _parse_tail(layers):
    """[BN, act, (conv, BN, act)*] -> (BatchNorms, convs, slopes), else None: the shape of tail the
    fused MFMA kernels take (ops.mlp_tail)."""
    bns, convs, slopes = [], [], []
    i, n = 0, len(layers)
    while i < n:
        if i > 0:
            if not isinstance(layers[i], nn.Conv2d) or layers[i].kernel_size != (1, 1):
                return None
            convs.append(layers[i])
            i += 1
        if i + 1 >= n or not isinstance(layers[i], (nn.BatchNorm2d, nn.BatchNorm1d)):
            return None
        bn, slope = layers[i], _act_slope(layers[i + 1])
        if slope is None or not (bn.training and bn.momentum is not None and bn.affine):
            return None
        bns.append(bn)
        slopes.append(slope)
        i += 2
    return (bns, convs, slopes) if convs else None


FUSED_TAILS = [True]      # ops.mlp_tail for bf16 tails (off: the separate BN / GEMM launches, for A/B runs)


def _try_fused_tail(bns, convs, slopes, x, K, nseg):
    """x (P,C0) rows -> (P/K, C_L) through ops.mlp_tail, or None when the tail is not one it takes."""
    if not (FUSED_TAILS[0] and x.is_cuda and x.dtype == torch.bfloat16 and rows_first()):
        return None
    chans = [x.shape[-1]] + [c.out_channels for c in convs]
    if not ops.mlp_tail_supported(x, chans, K) or any(c.in_channels != a for c, a in zip(convs, chans[:-1])):
        return None
    Ws = [(conv_weights_seg(c, nseg) if nseg > 1 else conv_weight2d(c)).float() for c in convs]
    return ops.mlp_tail(x, bns, Ws, slopes, K, nseg, shifts=[None] + [c.bias for c in convs])


def mlp_tail_rows(layers, x, reduce_max=False, nseg=1):
    """Run [conv, (bn), act]* layers (from a given position) on rows x (...,K,C).

    BatchNorm + activation pairs go through the fused kernel; with `reduce_max` the max over
    the second-to-last axis (the K neighbours) is fused into the final pair.
    nseg > 1: the leading axis holds nseg equal blocks (frames, fake / real batch) that the
    reference pushes through this tail one call after the other; here they share every launch --
    one batched GEMM with the nseg successive weights, one segmented BatchNorm pass -- with the
    statistics, running statistics and spectral-norm iterations of the separate calls."""
    lead, K = x.shape[:-2], x.shape[-2]
    if nseg > 1 and not _segmentable(layers, x, K):
        outs = [mlp_tail_rows(layers, xs, reduce_max) for xs in x.chunk(nseg, 0)]
        return torch.cat(outs, 0)
    x = x.reshape(-1, x.shape[-1])
    if reduce_max:
        parsed = _parse_tail(layers)
        if parsed is not None:
            out = _try_fused_tail(*parsed, x, K, nseg)
            if out is not None:
                return out.view(*lead, out.shape[-1])
    i, n, reduced, shift = 0, len(layers), False, None
    while i < n:
        m = layers[i]
        if isinstance(m, nn.Conv2d) and nseg > 1:
            x = rows_matmul_seg(x, conv_weights_seg(m, nseg))      # bias: folded into the next BN
            shift = m.bias
        elif isinstance(m, nn.Conv2d):
            x = rows_matmul(x, conv_weight2d(m), m.bias)
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            slope = _act_slope(layers[i + 1]) if i + 1 < n else None
            last = reduce_max and not any(isinstance(l, nn.Conv2d) for l in layers[i + 1:])
            if slope is not None and _fusable(x, K if last else 0):
                x = bn_act_rows(m, x, slope, K if last else 0, nseg=nseg, mean_shift=shift)
                shift = None
                reduced = reduced or last
                i += 1                                             # the activation is consumed
            else:
                assert nseg == 1
                x = bn_rows(m, x)
        else:
            x = m(x)
        i += 1
    if reduce_max:
        if not reduced:
            x = x.view(-1, K, x.shape[-1]).max(dim=1)[0]
        return x.view(*lead, x.shape[-1])
    return x.view(*lead, K, x.shape[-1])


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

    def sample_centres(self, xyz, prefix=None):
        """FPS centres of this level.  prefix: a one-element list holding the flag tensor of the sampling that produced
        xyz's ORDER (xyz = the previous level's centres, in pick order) or None; it is replaced by this level's flag
        for the level after (ops.furthest_point_sample_prefix: the sampling of an FPS prefix is 0..m-1, exactly)."""
        if prefix is not None and xyz.is_cuda and not (self.mask_dummy and self.check_dummies):
            centres, prefix[0] = ops.furthest_point_sample_prefix(xyz, self.npoint, prefix[0])
            return centres
        if prefix is not None:
            prefix[0] = None                     # (dummy replacement re-orders the centres: no shortcut downstream)
        centres = ops.furthest_point_sample(xyz, self.npoint)
        if self.mask_dummy and self.check_dummies:
            centres = replace_dummy_centres(xyz, centres)
        return centres

    def index_level(self, xyz, inverse=True, prefix=None):
        """All index-only work of this level for detached clouds xyz (B,N,3):
        -> (centres (B,S) int32, new_xyz (B,S,3) detached, idx (B,S,ns) int32).
        It depends on coordinates only, so a caller may run it ahead of time / on another stream
        (see `index_plan` of the discriminators) and hand the result to `forward_rows`.
        inverse=False leaves the inverted index to whoever regroups the lists (`merge_plans`).
        prefix: see `sample_centres` (pass the same list to every level of a chain, [None] at the first)."""
        xyz = xyz.detach().float().contiguous()
        centres = self.sample_centres(xyz, prefix)
        new_xyz = _gather_centres(xyz, centres)
        g = self.groupers[0]
        idx = _sources(ops.ball_query(g.radius, g.nsample, xyz, new_xyz), xyz.shape[1])
        if inverse and rows_first():
            ops.attach_inverse(idx, xyz.shape[1])       # for the backward of the row gather
        return centres, new_xyz, idx

    def _first_layer(self, grouper, mlp, xyz, new_xyz, feat_rows, idx=None):
        """Rows of the first conv's output for every grouped position: (B,S,ns,C1)."""
        conv = mlp[0]
        rows_dtype = amp_dtype(xyz)           # decided OUTSIDE the fp32 island below
        with no_autocast(xyz):
            W = conv_weight2d(conv).float()
            src = xyz if feat_rows is None else torch.cat([xyz, feat_rows.float()], dim=-1)
            if not grouper.use_xyz:
                src = feat_rows.float()
            if isinstance(grouper, GroupAll):
                return rows_matmul(src, W, conv.bias).unsqueeze(1)   # one group holding all N points
            U = rows_matmul(src, W, conv.bias)                      # (B,N,C1)
            if idx is None:
                idx = ops.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
            if grouper.use_xyz:
                Q = rows_matmul(new_xyz, W[:, :3])                  # centre term of (xyz_j - c_i)
                return ops.row_combine(U, Q, idx, ops.ROW_SUB, out_dtype=rows_dtype)
            return ops.row_combine(U, None, idx, ops.ROW_GATHER, out_dtype=rows_dtype)

    def forward_rows(self, xyz, feat_rows, plan=None):
        """xyz (B,N,3), feat_rows (B,N,C)|None -> new_xyz (B,npoint,3)|None, (B,npoint,C') rows.
        plan = (centres, idx) from `index_level` (single-scale levels only)."""
        xyz = xyz.float().contiguous()
        pidx = None
        if self.npoint is not None:
            if plan is not None:
                centres, pidx = plan
            else:
                centres = self.sample_centres(xyz)
            new_xyz = _gather_centres(xyz, centres)
        else:
            new_xyz = None
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            if rows_first():
                y = self._first_layer(grouper, mlp, xyz, new_xyz, feat_rows, pidx)
                outs.append(mlp_tail_rows(list(mlp)[1:], y, reduce_max=True))   # (B,S,C')
            else:                                                  # discriminator.py:139-150
                planes = None if feat_rows is None else feat_rows.float().transpose(1, 2).contiguous()
                g = mlp(grouper(xyz, new_xyz, planes))             # (B,C',S,ns)
                outs.append(F.max_pool2d(g, kernel_size=[1, g.size(3)]).squeeze(-1).transpose(1, 2))
        return new_xyz, torch.cat(outs, dim=-1)

    def frames_stackable(self):
        single = len(self.groupers) == 1 and isinstance(self.groupers[0], QueryAndGroup)
        return rows_first() and self.npoint is not None and single and self.groupers[0].use_xyz

    def forward_rows_frames(self, xyz_lst, feat_rows_lst, plan=None):
        """T frames of one clip through this level: lists of (B,N,3) / (B,N,C) -> lists.

        Exactly T separate `forward_rows` calls (per-frame spectral-norm power iterations,
        per-frame BatchNorm statistics), except that the T calls share their launches: the
        index-only work -- FPS, centre gather, ball query, the row gather -- runs ONCE on the
        T*B stacked clouds (no cross-cloud coupling; FPS in particular is a chain of npoint-1
        dependent rounds whose latency is paid per launch, not per cloud), and the MLP runs
        as T segments of batched GEMMs / segmented BatchNorm passes (`forward_rows_stacked`)."""
        T = len(xyz_lst)
        if T == 1 or not self.frames_stackable():
            assert plan is None, "index plans need the stacked-frames path"
            outs = [self.forward_rows(x, f) for x, f in zip(xyz_lst, feat_rows_lst)]
            return [o[0] for o in outs], [o[1] for o in outs]
        B = xyz_lst[0].shape[0]
        xyz = torch.cat([x.float() for x in xyz_lst], 0)
        feat = None if feat_rows_lst[0] is None else torch.cat(list(feat_rows_lst), 0)
        new_xyz, feats = self.forward_rows_stacked(xyz, feat, T, plan)
        return list(new_xyz.view(T, B, *new_xyz.shape[1:]).unbind(0)), list(feats.view(T, B, *feats.shape[1:]).unbind(0))

    def forward_rows_stacked(self, xyz, feat, nseg, plan=None):
        """`nseg` calls of this level in one pass: xyz (nseg*B,N,3), feat (nseg*B,N,C)|None, the
        calls stacked along the cloud axis in call order -> new_xyz (nseg*B,S,3), rows
        (nseg*B,S,C').  plan = (centres, idx) of the stacked clouds (`index_level`)."""
        assert self.frames_stackable()
        grouper, mlp = self.groupers[0], self.mlps[0]
        xyz = xyz.float().contiguous()
        NB, N, _ = xyz.shape
        centres = plan[0] if plan is not None else self.sample_centres(xyz)
        new_xyz = _gather_centres(xyz, centres)
        idx = plan[1] if plan is not None else ops.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
        S = new_xyz.shape[1]
        conv = mlp[0]
        rows_dtype = amp_dtype(xyz)
        with no_autocast(xyz):
            # first conv BEFORE the gather, per call with that call's weight: y = U[idx] - Q,
            # U = W_s [xyz|feat] (all points), Q = W_s[:, :3] centre - bias (bias folded into Q)
            W = conv_weights_seg(conv, nseg).float()                           # (nseg, C1, Cin)
            src = xyz if feat is None else torch.cat([xyz, feat.float()], dim=-1)
            U = rows_matmul_seg(src.view(NB * N, -1), W).view(NB, N, -1)
            Q = rows_matmul_seg(new_xyz.view(NB * S, 3), W[:, :, :3].contiguous()).view(NB, S, -1)
            if conv.bias is not None:
                Q = Q - conv.bias.float()
        y = ops.row_combine(U, Q, idx, ops.ROW_SUB, out_dtype=rows_dtype)        # (nseg*B, S, K, C1)
        feats = mlp_tail_rows(list(mlp)[1:], y, reduce_max=True, nseg=nseg)     # (nseg*B, S, C')
        return new_xyz, feats

    def forward_rows_pool_stacked(self, xyz, feat, nseg):
        """`nseg` calls of a GroupAll level (npoint None: one group holding all N points) in one
        pass: xyz (nseg*B,N,3), feat (nseg*B,N,C) -> (nseg*B, C')."""
        grouper, mlp = self.groupers[0], self.mlps[0]
        assert self.npoint is None and isinstance(grouper, GroupAll) and len(self.groupers) == 1
        NB, N, _ = xyz.shape
        conv = mlp[0]
        with no_autocast(xyz):
            W = conv_weights_seg(conv, nseg).float()
            src = torch.cat([xyz.float(), feat.float()], dim=-1) if grouper.use_xyz else feat.float()
            y = rows_matmul_seg(src.reshape(NB * N, -1), W)
            if conv.bias is not None:
                y = y + conv.bias.float()
        y = y.view(NB, 1, N, -1)                                                   # one group of N rows per cloud
        return mlp_tail_rows(list(mlp)[1:], y, reduce_max=True, nseg=nseg).view(NB, -1)

    def forward(self, xyz, features):
        """Reference signature: xyz (B,N,3), features (B,C,N) -> new_xyz, (B,C',npoint)."""
        rows = None if features is None else features.transpose(1, 2)
        new_xyz, out = self.forward_rows(xyz, rows)
        return new_xyz, out.transpose(1, 2).contiguous()


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

    def forward_rows(self, p1, p2, f1, f2, radius, idx=None):
        """p (B,N,3), f (B,N,C) rows -> (B,N,mlp[-1]) rows; idx = precomputed neighbour list.

        First conv on cat([pos2_j - pos1_i, feat2_j, feat1_i]) (discriminator.py:270-280) split
        by input columns: U = W[:, :3+C] [pos2|feat2] is gathered, Q = W[:, :3] pos1 - W[:, 3+C:] feat1
        is the per-centre term."""
        C = f1.shape[-1]
        if idx is None:
            idx = ball_query_wrapper(radius, self.NSAMPLE, p1, p2).to(torch.int32).contiguous()
        if not rows_first():                         # discriminator.py:270-283
            B, N, _ = p1.shape
            pos1, pos2 = p1.transpose(1, 2).contiguous(), p2.transpose(1, 2).contiguous()
            pos_diff = ops.grouping_operation(pos2.float(), idx) - pos1.view(B, -1, N, 1)
            feat2 = ops.grouping_operation(f2.float().transpose(1, 2).contiguous(), idx)
            feat1 = f1.transpose(1, 2).reshape(B, -1, N, 1).expand(-1, -1, -1, self.NSAMPLE)
            x = torch.cat([pos_diff, feat2, feat1], dim=1)
            for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                x = F.leaky_relu(bn(conv(x)))
            return torch.max(x, -1)[0].transpose(1, 2)
        with no_autocast(p1):
            W = conv_weight2d(self.mlp_convs[0]).float()
            U = rows_matmul(torch.cat([p2.float(), f2.float()], dim=-1), W[:, :3 + C])
            Q = rows_matmul(p1.float(), W[:, :3]) - rows_matmul(f1.float(), W[:, 3 + C:])
        x = ops.row_combine(U, Q, idx, ops.ROW_SUB, out_dtype=amp_dtype(p1))     # (B,N,32,C1)
        B, N, K, _ = x.shape
        x = x.view(B * N * K, -1)
        nl = len(self.mlp_convs)
        if all(bn.training and bn.momentum is not None for bn in self.mlp_bns):
            out = _try_fused_tail(list(self.mlp_bns), list(self.mlp_convs)[1:], [0.01] * nl, x, K, 1)
            if out is not None:
                return out.view(B, N, -1)
        for l in range(nl):                                        # F.leaky_relu default slope 0.01
            if l:
                x = rows_matmul(x, conv_weight2d(self.mlp_convs[l]))
            last = l == nl - 1
            if _fusable(x, K if last else 0):
                x = bn_act_rows(self.mlp_bns[l], x, 0.01, K if last else 0)
            else:
                x = F.leaky_relu(bn_rows(self.mlp_bns[l], x))
                if last:
                    x = x.view(B * N, K, -1).max(dim=1)[0]
        return x.view(B, N, -1)

    def forward_rows_stacked(self, p1, p2, f1, f2, idx, nseg):
        """`nseg` calls of this layer in one pass (calls stacked along the cloud axis, in call
        order): p (nseg*B,N,3), f (nseg*B,N,C), idx (nseg*B,N,32) -> (nseg*B,N,mlp[-1])."""
        C = f1.shape[-1]
        NB, N, _ = p1.shape
        with no_autocast(p1):
            W = conv_weights_seg(self.mlp_convs[0], nseg).float()              # (nseg, C1, 3+2C)
            U = rows_matmul_seg(torch.cat([p2.float(), f2.float()], dim=-1).view(NB * N, -1),
                                W[:, :, :3 + C].contiguous()).view(NB, N, -1)
            Q = (rows_matmul_seg(p1.float().reshape(NB * N, 3), W[:, :, :3].contiguous())
                 - rows_matmul_seg(f1.float().reshape(NB * N, C), W[:, :, 3 + C:].contiguous())).view(NB, N, -1)
        x = ops.row_combine(U, Q, idx, ops.ROW_SUB, out_dtype=amp_dtype(p1))     # (nseg*B,N,32,C1)
        K = x.shape[2]
        x = x.view(NB * N * K, -1)
        nl = len(self.mlp_convs)
        if all(bn.training and bn.momentum is not None for bn in self.mlp_bns):
            out = _try_fused_tail(list(self.mlp_bns), list(self.mlp_convs)[1:], [0.01] * nl, x, K, nseg)
            if out is not None:
                return out.view(NB, N, -1)
        for l in range(nl):                                        # F.leaky_relu default slope 0.01
            if l:
                x = rows_matmul_seg(x, conv_weights_seg(self.mlp_convs[l], nseg))
            x = bn_act_rows(self.mlp_bns[l], x, 0.01, K if l == nl - 1 else 0, nseg=nseg)
        return x.view(NB, N, -1)

    def forward(self, pos1, pos2, feature1, feature2, radius):
        """Reference signature: pos (B,3,N), feature (B,C,N) -> pos1, (B,mlp[-1],N)."""
        out = self.forward_rows(pos1.transpose(1, 2).contiguous(), pos2.transpose(1, 2).contiguous(),
                                feature1.transpose(1, 2), feature2.transpose(1, 2), radius)
        return pos1, out.transpose(1, 2).contiguous()


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

    def pair_indices(self, pos_rows_lst, cutoff, inverse=True):
        """Neighbour lists of the frame pairs (l, l+1): positions only, shared by every depth."""
        pairs = [_sources(ball_query_wrapper(cutoff, FlowEmbedding.NSAMPLE, pos_rows_lst[l].detach(),
                                             pos_rows_lst[l + 1].detach()).to(torch.int32).contiguous(),
                          pos_rows_lst[l + 1].shape[1])
                 for l in range(len(pos_rows_lst) - 1)]
        if inverse and rows_first():
            for l, idx in enumerate(pairs):               # for the backward of the row gather
                ops.attach_inverse(idx, pos_rows_lst[l + 1].shape[1])
        return pairs

    def forward_rows(self, feat_rows_lst, pos_rows_lst, cutoff, pair_idx=None):
        """Lists of (B,N,C) / (B,N,3) rows -> (B,N,out) rows."""
        assert len(feat_rows_lst) == self.depth + 1
        feats = list(feat_rows_lst)
        if pair_idx is None and rows_first():
            pair_idx = self.pair_indices(pos_rows_lst, cutoff)     # depth d re-uses pairs 0..T-2-d
        for depth in range(self.depth):
            layer = self.flow_emb_layers[depth]
            feats = [layer.forward_rows(pos_rows_lst[l], pos_rows_lst[l + 1], feats[l], feats[l + 1], cutoff,
                                        None if pair_idx is None else pair_idx[l])
                     for l in range(len(feats) - 1)]
        assert len(feats) == 1
        return feats[0]

    def depth_indices(self, pair_idx_per_pass, n_src):
        """Neighbour lists of `forward_rows_passes`: for depth d the pairs 0..T-2-d of every pass,
        stacked pass-major along the cloud axis (with their inverted index, ops.attach_inverse)."""
        T1 = len(pair_idx_per_pass[0])
        out = []
        for d in range(self.depth):
            idx = torch.cat([pairs[l] for pairs in pair_idx_per_pass for l in range(T1 - d)], 0).contiguous()
            out.append(ops.attach_inverse(idx, n_src))
        return out

    def forward_rows_passes(self, feats, poss, depth_idx):
        """NP passes x T frames at once: feats (NP,T,B,N,C), poss (NP,T,B,N,3) -> (NP*B,N,out).
        Every depth runs its NP*(T-1-d) calls as segments of one pass (pass-major call order, the
        order NP successive forward_rows calls would make them in)."""
        NP, T = feats.shape[:2]
        assert T == self.depth + 1
        for d in range(self.depth):
            n = T - 1 - d                                           # calls per pass at this depth
            flat = lambda t: t.reshape(-1, *t.shape[3:])            # noqa: E731  (NP,n,B,..) -> (NP*n*B,..)
            out = self.flow_emb_layers[d].forward_rows_stacked(
                flat(poss[:, :n]), flat(poss[:, 1:n + 1]), flat(feats[:, :n]), flat(feats[:, 1:n + 1]),
                depth_idx[d], NP * n)
            feats = out.view(NP, n, out.shape[0] // (NP * n), *out.shape[1:])
        return feats.reshape(-1, *feats.shape[3:])                  # n == 1 at the last depth

    def forward(self, feature_lst, pos_lst, cutoff):
        """Reference signature: lists of (B,C,N) / (B,3,N) -> (B,out,N)."""
        out = self.forward_rows([f.transpose(1, 2) for f in feature_lst],
                                [p.transpose(1, 2).contiguous() for p in pos_lst], cutoff)
        return out.transpose(1, 2).contiguous()


def _head(dims, drops):
    """sn-Linear -> BN1d -> LeakyReLU [-> Dropout] ... -> sn-Linear(.,1), reference indices kept."""
    layers = []
    for i in range(len(dims) - 2):
        layers += [spectral_norm(nn.Linear(dims[i], dims[i + 1])), nn.BatchNorm1d(dims[i + 1]), nn.LeakyReLU()]
        if drops[i]:
            layers.append(nn.Dropout(drops[i]))
    layers.append(spectral_norm(nn.Linear(dims[-2], dims[-1])))
    return nn.Sequential(*layers)


FUSED_HEAD = [os.environ.get("TPGAN_FUSED_HEAD", "1") != "0"]     # off: PyTorch's BatchNorm1d / LeakyReLU / Dropout (A/B runs)
_ONES = {}


def _ones_like(x):
    """A constant tensor of ones of x's shape (kept: a fill launch per use otherwise; never written)."""
    key = (tuple(x.shape), x.device, x.dtype)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones_like(x)
    return t


def _head_fp32(fc_layers, x):
    """The (B,C) classification head, always in fp32 (B rows: nothing to gain from bf16);
    spectrally-normalised linears use the fused kernel, everything else its own module."""
    with no_autocast(x):
        x = x.float()
        if not rows_first() or not isinstance(fc_layers, nn.Sequential):
            return fc_layers(x)                       # reference order: PyTorch's own hooks
        mods, i = list(fc_layers), 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                x = rows_matmul(x, sn_weight(m), m.bias)
            elif (FUSED_HEAD[0] and isinstance(m, nn.BatchNorm1d) and m.training and m.momentum is not None
                  and m.track_running_stats and x.dim() == 2 and x.shape[0] > 1 and i + 1 < len(mods)
                  and isinstance(mods[i + 1], nn.LeakyReLU)):
                # BatchNorm1d -> LeakyReLU [-> Dropout] in ONE launch each way (ops.head_bn_act, csrc/head.hip); the
                # dropout's scaled keep mask is drawn by the module itself on a constant tensor of ones -- the same
                # draws, in the same order, as its call on the activations
                act, drop, mask = mods[i + 1], None, None
                if i + 2 < len(mods) and isinstance(mods[i + 2], nn.Dropout):
                    drop = mods[i + 2]
                    if drop.training and drop.p > 0.0:
                        mask = drop(_ones_like(x))
                x = ops.head_bn_act(x, m, act.negative_slope, mask)
                i += 1 if drop is None else 2
            else:
                x = m(x)
            i += 1
        return x


def _gather_centres(xyz, centres):
    """Coordinates of the sampled centres, (B,S,3) rows: gather_operation (discriminator.py:131-137) -- on the rows path as
    one launch on the rows themselves, in the reference's order through the (B,3,N) planes it uses."""
    if rows_first():
        return ops.gather_rows(xyz.contiguous(), centres)
    return ops.gather_operation(xyz.transpose(1, 2).contiguous(), centres).transpose(1, 2).contiguous()


def _sources(idx, n_src):
    """Remember on a neighbour list how many source rows it indexes (merge_plans needs it)."""
    idx._tpg_nsrc = int(n_src)
    return idx


def _cut(idx, lo, hi):
    """Clouds lo..hi of a neighbour list (a view: the clouds are the leading, contiguous axis)."""
    return _sources(idx[lo:hi], idx._tpg_nsrc)


def attach_plan_inverses(plan):
    """Give every ball-query list of a per-pass plan of `index_plans` its inverted index, so the
    plan can go to `forward(..., plan=)` directly instead of through `merge_plans`."""
    for _, idx in plan["sa"]:
        ops.attach_inverse(idx, idx._tpg_nsrc)
    for idx in plan.get("flow", []):
        ops.attach_inverse(idx, idx._tpg_nsrc)
    return plan


def run_index_plan(make_plan, stream):
    """Run `make_plan()` (an `index_plan` call, possibly preceded by cheap tensor prep) on
    `stream`, forked from the current stream; returns (result, join) where `join()` makes the
    current stream wait for exactly this piece of work.  The FPS chain is m-1 dependent rounds
    on B workgroups: on a 256-CU part it costs latency, not throughput, so it is issued early and
    overlapped with whatever the main stream is doing.  Works eagerly and inside hipGraph
    capture (fork / join become graph dependencies)."""
    main = torch.cuda.current_stream(stream.device)
    stream.wait_stream(main)
    with torch.cuda.stream(stream):
        result = make_plan()
        done = torch.cuda.Event()
        done.record(stream)
    for t in _plan_tensors(result):
        t.record_stream(main)

    def join():
        torch.cuda.current_stream(stream.device).wait_event(done)
    return result, join


def _plan_tensors(obj):
    """Every tensor inside a (nested) plan / tuple / list / dict."""
    if torch.is_tensor(obj):
        inv = getattr(obj, "_tpg_inverse", None)         # ops.attach_inverse
        return [obj] + ([inv[1], inv[2]] if inv is not None else [])
    if isinstance(obj, dict):
        obj = list(obj.values())
    if isinstance(obj, (list, tuple)):
        return [t for o in obj for t in _plan_tensors(o)]
    return []


class _TempoDis(nn.Module):
    """Per-frame SA x2 -> FlowModule over the T frames -> GroupAll SA -> FC head."""

    flow_radius_scale = 1.0

    def index_plan(self, pos_lst, cutoff):
        """Every index of one forward over the frames `pos_lst` -- FPS centres and ball-query
        lists of both levels (frames stacked) and the flow-embedding neighbour lists.  Depends on
        coordinates only; see `run_index_plan` for running it on a side stream."""
        T, B = len(pos_lst), pos_lst[0].shape[0]
        xyz = torch.cat([p.detach().float() for p in pos_lst], 0)
        chain = [None]                              # level 1 samples level 0's centres: FPS-prefix shortcut
        c0, x1, i0 = self.coarse_graining_module[0].index_level(xyz, prefix=chain)
        c1, x2, i1 = self.coarse_graining_module[1].index_level(x1, prefix=chain)
        pairs = self.flow_module.pair_indices([x2[t * B:(t + 1) * B] for t in range(T)],
                                              self.flow_radius_scale * cutoff)
        return {"sa": [(c0, i0), (c1, i1)], "flow": pairs}

    def index_plans(self, pos_lsts, cutoff):
        """[index_plan(p, cutoff) for p in pos_lsts] (same shapes) with every search launched ONCE
        for all passes: furthest point sampling is npoint-1 dependent rounds on one workgroup per
        cloud, so on a 256-CU part the passes' clouds cost the rounds of one pass.  Every search is
        per cloud, so the lists are the ones separate calls give.  The per-pass plans come without
        inverted indices: hand them to `merge_plans` (or `attach_plan_inverses`)."""
        NP, T, B = len(pos_lsts), len(pos_lsts[0]), pos_lsts[0][0].shape[0]
        xyz = torch.cat([p.detach().float() for pos_lst in pos_lsts for p in pos_lst], 0)   # pass-major
        chain = [None]
        c0, x1, i0 = self.coarse_graining_module[0].index_level(xyz, inverse=False, prefix=chain)
        c1, x2, i1 = self.coarse_graining_module[1].index_level(x1, inverse=False, prefix=chain)
        frames = x2.view(NP, T, B, *x2.shape[1:])
        pairs = self.flow_module.pair_indices([frames[:, t].reshape(NP * B, *x2.shape[1:]) for t in range(T)],
                                              self.flow_radius_scale * cutoff, inverse=False)
        n = T * B
        return [{"sa": [(c0[p * n:(p + 1) * n], _cut(i0, p * n, (p + 1) * n)),
                        (c1[p * n:(p + 1) * n], _cut(i1, p * n, (p + 1) * n))],
                 "flow": [_cut(pr, p * B, (p + 1) * B) for pr in pairs]} for p in range(NP)]

    def _levels(self, pos_lst, feat_lst, plan=None):
        feats0 = list(feat_lst) if feat_lst is not None else list(pos_lst)
        sa = plan["sa"] if plan is not None else (None, None)
        poss, feats = self.coarse_graining_module[0].forward_rows_frames(list(pos_lst), feats0, sa[0])
        poss2, feats2 = self.coarse_graining_module[1].forward_rows_frames(poss, feats, sa[1])
        return feats2, poss2

    def _sn_calls(self, T, passes=1):
        """(module, calls in `passes` forwards over T frames) for every conv / linear."""
        calls = []
        for sa in self.coarse_graining_module:
            calls += [(m, T * passes) for mlp in sa.mlps for m in mlp if isinstance(m, nn.Conv2d)]
        for d, layer in enumerate(self.flow_module.flow_emb_layers):
            calls += [(m, (T - 1 - d) * passes) for m in layer.mlp_convs]
        calls += [(m, passes) for mlp in self.SA_pooling.mlps for m in mlp if isinstance(m, nn.Conv2d)]
        calls += [(m, passes) for m in self.fc_layers.modules() if isinstance(m, nn.Linear)]
        return calls

    def prepare_sn(self, T, passes=1):
        """Spectral-norm weights of a later forward / forward_passes over T frames, computed now
        (sn_prepare)."""
        sn_prepare(self._sn_calls(T, passes), self.training)

    def merge_plans(self, plans):
        """Index plans of successive forwards (same shapes) -> the plan of `forward_passes`."""
        n0 = plans[0]["sa"][0][1]._tpg_nsrc
        n1 = plans[0]["sa"][1][1]._tpg_nsrc
        sa = []
        for l, n_src in ((0, n0), (1, n1)):
            c = torch.cat([p["sa"][l][0] for p in plans], 0)
            i = ops.attach_inverse(torch.cat([p["sa"][l][1] for p in plans], 0).contiguous(), n_src)
            sa.append((c, i))
        n2 = plans[0]["flow"][0]._tpg_nsrc
        return {"sa": sa, "flow_depth": self.flow_module.depth_indices([p["flow"] for p in plans], n2)}

    def forward_passes(self, pos_lsts, cutoff, plan=None):
        """[forward(pos_lst, cutoff) for pos_lst in pos_lsts] -- the same module calls in the same
        order (so: the same spectral-norm iterations, BatchNorm statistics and running-statistic
        updates) -- with the passes sharing their launches: every level, flow-embedding depth and
        the pooling level run all passes (x frames) as segments of one batched pass.  Used for the
        fake and the real batch of a discriminator update (train_step_final.py:171-204).
        plan = merge_plans([index_plan(p, cutoff) for p in pos_lsts]) or None."""
        NP, T, B = len(pos_lsts), len(pos_lsts[0]), pos_lsts[0][0].shape[0]
        lvl0, lvl1 = self.coarse_graining_module
        if not (rows_first() and lvl0.frames_stackable() and lvl1.frames_stackable() and pos_lsts[0][0].is_cuda):
            assert plan is None
            return [self.forward(p, cutoff) for p in pos_lsts]
        with sn_prefetch(self._sn_calls(T, NP), self.training):
            xyz = torch.cat([p.float() for pos_lst in pos_lsts for p in pos_lst], 0)        # (NP*T*B, N, 3)
            if plan is None:
                plan = self.merge_plans([self.index_plan(p, cutoff) for p in pos_lsts])
            x1, f1 = lvl0.forward_rows_stacked(xyz, xyz, NP * T, plan["sa"][0])
            x2, f2 = lvl1.forward_rows_stacked(x1, f1, NP * T, plan["sa"][1])
            poss = x2.view(NP, T, B, *x2.shape[1:])
            f = self.flow_module.forward_rows_passes(f2.view(NP, T, B, *f2.shape[1:]), poss, plan["flow_depth"])
            pooled = self.SA_pooling.forward_rows_pool_stacked(poss[:, 0].reshape(NP * B, *x2.shape[1:]), f, NP)
            return [_head_fp32(self.fc_layers, h) for h in pooled.view(NP, B, -1).unbind(0)]

    def _forward(self, pos_lst, cutoff, feat_lst, width, plan=None):
        if feat_lst is not None:
            assert len(feat_lst) == len(pos_lst)
        with sn_prefetch(self._sn_calls(len(pos_lst)), self.training):
            feats, poss = self._levels(pos_lst, feat_lst, plan)     # rows all the way
            f = self.flow_module.forward_rows(feats, poss, self.flow_radius_scale * cutoff,
                                              None if plan is None else plan["flow"])
            _, f = self.SA_pooling.forward_rows(poss[0], f)
            return _head_fp32(self.fc_layers, f.reshape(-1, width))


class ActionTempoDis(_TempoDis):
    def __init__(self, sequence_length, sn=True):
        super().__init__()
        self.coarse_graining_module = nn.ModuleList([
            SSGSetConv(npoint=512, radius=0.8, nsample=64, mlp=[3, 64, 64, 128], use_xyz=True, sn=sn),
            SSGSetConv(npoint=256, radius=1.2, nsample=32, mlp=[128, 128, 256], use_xyz=True, sn=sn)])
        self.flow_module = FlowModule(256, 256, 256, sequence_length, sn=sn)
        self.SA_pooling = SSGSetConv(mlp=[256, 256, 512], use_xyz=True, sn=sn)
        self.fc_layers = _head([512, 256, 64, 1], [0.3, 0.1])

    def forward(self, pos_lst, cutoff, plan=None):
        return self._forward(pos_lst, cutoff, None, 512, plan)


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

    def forward(self, pos_lst, cutoff, feat_lst=None, plan=None):
        return self._forward(pos_lst, cutoff, feat_lst, 256, plan)


class _SpatialDis(nn.Module):
    def prepare_sn(self, passes=1):
        """Spectral-norm weights of a later forward / forward_passes, computed now (sn_prepare)."""
        sn_prepare(self._sn_calls(passes), self.training)

    def _sn_calls(self, passes=1):
        calls = []
        for sa in list(self.coarse_graining_module) + [self.SA_pooling]:
            calls += [(m, passes) for mlp in sa.mlps for m in mlp if isinstance(m, nn.Conv2d)]
        calls += [(m, passes) for m in self.fc_layers.modules() if isinstance(m, nn.Linear)]
        return calls

    def merge_plans(self, plans):
        """Index plans of successive forwards (same shapes) -> the plan of `forward_passes`."""
        sa = []
        for l in range(len(self.coarse_graining_module)):
            n_src = plans[0]["sa"][l][1]._tpg_nsrc
            c = torch.cat([p["sa"][l][0] for p in plans], 0)
            sa.append((c, ops.attach_inverse(torch.cat([p["sa"][l][1] for p in plans], 0).contiguous(), n_src)))
        return {"sa": sa}

    def forward_passes(self, pos_list, plan=None):
        """[forward(pos) for pos in pos_list] with the passes sharing their launches (see
        _TempoDis.forward_passes)."""
        NP, B = len(pos_list), pos_list[0].shape[0]
        if not (rows_first() and pos_list[0].is_cuda and all(sa.frames_stackable() for sa in self.coarse_graining_module)):
            assert plan is None
            return [self.forward(p) for p in pos_list]
        with sn_prefetch(self._sn_calls(NP), self.training):
            pos = torch.cat([p.float() for p in pos_list], 0)
            if plan is None:
                plan = self.merge_plans([self.index_plan(p) for p in pos_list])
            feature = None
            for l, sa in enumerate(self.coarse_graining_module):
                pos, feature = sa.forward_rows_stacked(pos, pos if feature is None else feature, NP, plan["sa"][l])
            pooled = self.SA_pooling.forward_rows_pool_stacked(pos, feature, NP)
            return [_head_fp32(self.fc_layers, h) for h in pooled.view(NP, B, -1).unbind(0)]

    def index_plans(self, pos_list):
        """[index_plan(p) for p in pos_list] with every search launched once for all passes (see
        _TempoDis.index_plans); per-pass plans without inverted indices."""
        NP, B = len(pos_list), pos_list[0].shape[0]
        xyz, levels, chain = torch.cat([p.detach().float() for p in pos_list], 0), [], [None]
        for sa in self.coarse_graining_module:
            c, xyz, i = sa.index_level(xyz, inverse=False, prefix=chain)
            levels.append((c, i))
        return [{"sa": [(c[p * B:(p + 1) * B], _cut(i, p * B, (p + 1) * B)) for c, i in levels]} for p in range(NP)]

    def index_plan(self, pos):
        """FPS centres + ball-query lists of every level for the clouds `pos` (coordinates only)."""
        xyz, levels, chain = pos.detach().float(), [], [None]
        for sa in self.coarse_graining_module:
            c, xyz, i = sa.index_level(xyz, prefix=chain)
            levels.append((c, i))
        return {"sa": levels}

    def _forward(self, pos, width, plan=None):
        with sn_prefetch(self._sn_calls(), self.training):
            feature = None
            for l, sa in enumerate(self.coarse_graining_module):
                pos, feature = sa.forward_rows(pos, pos if feature is None else feature,
                                               None if plan is None else plan["sa"][l])
            _, feature = self.SA_pooling.forward_rows(pos, feature)
            return _head_fp32(self.fc_layers, feature.reshape(-1, width))


class ActionSpatialDis(_SpatialDis):
    def __init__(self, sn=True):
        super().__init__()
        self.coarse_graining_module = nn.ModuleList([
            SSGSetConv(npoint=512, radius=0.3, nsample=32, mlp=[3, 64, 64, 128], use_xyz=True, sn=sn),
            SSGSetConv(npoint=256, radius=0.6, nsample=32, mlp=[128, 128, 128], use_xyz=True, sn=sn),
            SSGSetConv(npoint=128, radius=1.0, nsample=32, mlp=[128, 128, 256], use_xyz=True, sn=sn)])
        self.SA_pooling = SSGSetConv(mlp=[256, 256, 512], use_xyz=True, sn=sn)
        self.fc_layers = _head([512, 256, 64, 1], [0.3, 0.1])

    def forward(self, pos, plan=None):
        return self._forward(pos, 512, plan)


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

    def forward(self, pos, plan=None):
        return self._forward(pos, 256, plan)
