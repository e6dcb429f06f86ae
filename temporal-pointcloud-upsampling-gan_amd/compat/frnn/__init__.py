"""``frnn.frnn_grid_points`` served by the MI355X HIP kernels.

Reference call sites: discriminator.py:27-32; loss.py:105,142,229,256-265;
gcn_lib/interpolation.py:20,33; gcn_lib/pointnet/gcn.py:30 (4-tuple unpacked).
For the cloud sizes on this path (<= a few 10^4 points) an exhaustive wave-per-query
search beats building a uniform grid, so `grid` is accepted/returned as None."""
import torch

import tpgan_amd.ops as _ops


def frnn_gather(x, idx, lengths=None):
    """x (B,M,U), idx (B,L,K) with -1 = missing -> (B,L,K,U) with zeros at missing slots."""
    safe = idx.clamp(min=0)
    U = x.shape[2]
    K = idx.shape[2]
    out = torch.gather(x[:, :, None].expand(-1, -1, K, -1), 1, safe[:, :, :, None].expand(-1, -1, -1, U))
    return out.masked_fill((idx < 0)[..., None].expand_as(out), 0.0)


def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=-1, r=-1, grid=None,
                     return_nn=False, return_sorted=True, radius_cell_ratio=2.0):
    """<= K nearest points2 within distance r (d^2 < r^2) of each points1 row, ascending;
    missing slots are idx -1 / dist -1.  Returns (dists, idxs, nn|None, grid=None)."""
    if points1.shape[0] != points2.shape[0]:
        raise ValueError("points1 and points2 must have the same batch dimension")
    if points1.shape[2] != points2.shape[2]:
        raise ValueError("dimension mismatch")
    if K < 1 or K > 64:
        raise ValueError("K must be in [1, 64]")
    if isinstance(r, torch.Tensor):
        if r.numel() != 1:
            raise ValueError("per-cloud radii are not supported; pass one float")
        r = float(r)
    if r <= 0:
        raise ValueError("r must be positive")
    dists, idxs = _ops.neighbour_search(points1, points2, K, lengths1, lengths2, r=float(r))
    nn = frnn_gather(points2, idxs) if return_nn else None
    return dists, idxs, nn, None
