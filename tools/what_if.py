"""Critical-path sensitivity of the graph-replayed step: replace ONE search op by a free stand-in
(indices of the right shape, wrong values) and time the step.  The difference to the real step is the
most any optimisation of that op could buy; it is a measurement aid only, nothing here is product code.

    python tools/what_if.py [fps] [knn] [ball_query]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                             # noqa: E402

import bench                                                             # noqa: E402
from tpgan_amd import ops                                                # noqa: E402


def free_fps(self, xyz, m, start=None, skip_origin=True):
    B, N, _ = xyz.shape
    step = max(N // m, 1)
    return (torch.arange(m, device=xyz.device, dtype=torch.int32) * step % N).expand(B, m).contiguous()


_cache = {}


def free_knn(self, p1, p2, len1, len2, K, r2):
    B, P1, D = p1.shape
    key = (B, P1, p2.shape[1], K)
    if key not in _cache:
        idx = (torch.arange(P1, device=p1.device).view(1, P1, 1) + torch.arange(K, device=p1.device).view(1, 1, K) * 7) % p2.shape[1]
        _cache[key] = (torch.full((B, P1, K), 0.01, device=p1.device), idx.expand(B, P1, K).contiguous())
    return _cache[key]


def free_bq(self, radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    key = ("bq", B, N, S, nsample)
    if key not in _cache:
        idx = (torch.arange(S, device=xyz.device).view(1, S, 1) * (N // S) + torch.arange(nsample, device=xyz.device).view(1, 1, nsample)) % N
        _cache[key] = idx.to(torch.int32).expand(B, S, nsample).contiguous()
    return _cache[key]


def main():
    which = sys.argv[1:]
    if "fps" in which:
        ops.HipBackend.fps = free_fps
    if "knn" in which:
        ops.HipBackend.knn = free_knn
    if "ball_query" in which:
        ops.HipBackend.ball_query = free_bq
    sys.argv = [sys.argv[0], "--steps", "30", "--warmup", "3", "--no-extra"]
    print("stand-ins:", which or "none")
    bench.main()


if __name__ == "__main__":
    main()
