"""ball_query alone on the step's shapes (hipGraph replay of 20 launches): time per launch and the
algorithmic GB/s (12 B (N+S) + 4 B S nsample per cloud).  GPU box.

    python tools/tune_ballquery.py
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


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3


for B, N, S, r, ns in ((48, 4096, 1024, 0.10, 32), (24, 4096, 1024, 0.10, 32), (16, 4096, 1024, 0.15, 32),
                       (16, 1024, 512, 0.30, 32), (48, 1024, 256, 0.20, 32), (16, 512, 128, 0.60, 16),
                       (16, 16384, 1024, 0.10, 32), (40, 16384, 1024, 0.10, 32)):
    _, high = fluid_clip(B, N, 8, 1, seed=1, device=dev)
    x = high[0]
    fi = hip.fps(x, S)
    c = torch.gather(x, 1, fi.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    us = timeit(lambda: hip.ball_query(r, ns, x, c))
    nbytes = 12 * B * (N + S) + 4 * B * S * ns
    print(f"ball_query B={B:3d} {N:5d}->{S:4d} r={r:.2f} ns={ns:2d}: {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s")
