"""Feature-space kNN on large clouds: the matrix-core filter (csrc/knn_mfma.hpp) against the exhaustive kernels
(TPG_KNN_MFMA=0 in a second process), the share of queries it leaves to the exhaustive fallback, and the plain 3-D
kNN on the uniform grid against the exhaustive 3-D kernel.  GPU box.

    python tools/tune_knn_mfma.py            # filter + grid
    TPG_KNN_MFMA=0 python tools/tune_knn_mfma.py exhaustive
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
exhaustive = len(sys.argv) > 1 and sys.argv[1] == "exhaustive"


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def manifold(B, P, D, m):
    z = torch.randn(B, P, m, device=dev)
    return (torch.tanh(z @ torch.randn(m, D, device=dev)) + 0.05 * torch.randn(B, P, D, device=dev)).contiguous()


torch.manual_seed(0)
for (B, P, D, K) in ((40, 4096, 32, 20), (40, 4096, 64, 12), (24, 2048, 32, 20), (1, 16384, 32, 20), (1, 65536, 32, 20),
                     (1, 65536, 64, 12), (1, 100000, 64, 12)):
    for kind in ("normal", "manifold3", "manifold2"):
        x = torch.randn(B, P, D, device=dev) if kind == "normal" else manifold(B, P, D, int(kind[-1]))
        if exhaustive:
            print(f"kNN B={B} P={P} D={D} K={K} {kind:9s}: exhaustive {t(lambda: hip.knn(x, x, None, None, K, None), 3):9.1f} us", flush=True)
            continue
        us = t(lambda: hip.knn(x, x, None, None, K, None))
        _, raw = hip.knn_mfma(x, x, None, None, K, redo=False)
        print(f"kNN B={B} P={P} D={D} K={K} {kind:9s}: filter + fallback {us:9.1f} us, filter alone "
              f"{t(lambda: hip.knn_mfma(x, x, None, None, K, redo=False)):9.1f} us, "
              f"{(raw[:, :, 0] == -2).float().mean().item() * 100:6.2f} % of the queries to the fallback", flush=True)
if not exhaustive:
    for (B, N, K) in ((40, 4096, 20), (24, 2048, 20), (8, 16384, 1), (8, 16384, 20), (1, 65536, 20)):
        _, high = fluid_clip(B, N, 8, 1, seed=1, device=dev)
        x = high[0]
        res = {}
        for name, mn in (("exhaustive", 10 ** 9), ("grid", 1)):
            old = hip.KNN_GRID_MIN_POINTS, hip.GRID_MIN_PAIRS
            hip.KNN_GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = mn, 0.0
            try:
                res[name] = t(lambda: hip.knn(x, x, None, None, K, None))
            finally:
                hip.KNN_GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = old
        print(f"3-D kNN B={B} N={N} K={K}: exhaustive {res['exhaustive']:9.1f} us   grid {res['grid']:9.1f} us", flush=True)
