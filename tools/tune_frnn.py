"""Exhaustive wave-per-query radius search (knn.hip) against the uniform grid (frnn_grid.hip): time per call at
the cloud sizes of BASELINE cfg2 / cfg5 / the rollout, to place HipBackend.GRID_MIN_POINTS.  GPU box.

    python tools/tune_frnn.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import ops
from tpgan_amd.synthetic import fluid_clip

dev = torch.device("cuda", 0)
hip = ops.backend_for(torch.zeros(1, device=dev))


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3


for B, N in ((8, 1024), (8, 2048), (8, 4096), (8, 8192), (8, 16384), (1, 65536), (1, 131072)):
    _, high = fluid_clip(B, N, 8, 1, seed=1, device=dev)
    x = high[0]
    lowq = x[:, ::8].contiguous()
    for label, q, K, r in (("mask loss K=16 r=0.035 self", x, 16, 0.035), ("mask loss K=1 r=0.0475 low->high", lowq, 1, 0.0475)):
        res = {}
        for name, thr in (("exhaustive", 1 << 30), ("grid", 1)):
            hip.GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = thr, 0.0
            if name == "exhaustive" and N * q.shape[1] * B > 3e10:
                res[name] = float("nan")
                continue
            res[name] = timeit(lambda: hip.knn(q, x, None, None, K, ops.radius_sq(r), r=r))
        print(f"B={B} N={N:6d} {label:34s} exhaustive {res['exhaustive']:9.1f} us   grid {res['grid']:8.1f} us")
