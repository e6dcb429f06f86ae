"""``pytorch3d.ops.knn_points`` served by the MI355X HIP kernels.

Reference call sites: gcn_lib/pointnet/gcn.py:16-21,38; discriminator.py:15-20,33-38
(results are unpacked positionally as `dist, idx, _`)."""
from collections import namedtuple

import torch

import tpgan_amd.ops as _ops

_KNN = namedtuple("KNN", "dists idx knn")


def knn_gather(x, idx, lengths=None):
    """x (B,M,U), idx (B,L,K) -> (B,L,K,U); slots beyond `lengths` are zeroed."""
    B, M, U = x.shape
    _, L, K = idx.shape
    out = torch.gather(x[:, :, None].expand(-1, -1, K, -1), 1, idx[:, :, :, None].expand(-1, -1, -1, U))
    if lengths is not None and int(lengths.min()) < K:
        mask = lengths[:, None] <= torch.arange(K, device=x.device)[None]
        out = out.masked_fill(mask[:, None, :, None].expand(-1, L, -1, U), 0.0)
    return out


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False,
               return_sorted=True):
    """K nearest neighbours of each p1 row among p2 rows, squared L2, ascending (dist, idx).

    `version` is accepted and ignored (one kernel design); results are always sorted."""
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    dists, idx = _ops.neighbour_search(p1, p2, K, lengths1, lengths2, r=None)
    dists = _ops.attach_dist_grad(p1, p2, dists, idx)
    nn = None
    if return_nn:
        l2 = lengths2 if lengths2 is None else torch.as_tensor(lengths2, device=p1.device)
        nn = knn_gather(p2, idx, l2)
    return _KNN(dists=dists, idx=idx, knn=nn)
