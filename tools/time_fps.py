"""Furthest point sampling, us per launch and per round at the step's shapes (TPG_FPS_PRUNE=0: the dense rounds).  GPU box.

    python tools/time_fps.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import ops
from tpgan_amd.synthetic import fluid_clip, action_clip

dev = torch.device("cuda", 0)
hip = ops.backend_for(torch.zeros(1, device=dev))
for name, B, N, m in (("cfg2 fake+update clouds", 48, 4096, 1024), ("cfg2 real clouds", 24, 4096, 1024),
                      ("cfg5 shard", 40, 16384, 4096), ("cfg4", 64, 2048, 512), ("8192", 16, 8192, 2048)):
    _, hi = fluid_clip(B, N, 4, 1, seed=3, device=dev)
    x = hi[0].contiguous()
    for _ in range(2):
        hip.fps(x, m)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        hip.fps(x, m)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 * 1e3
    print(f"{name:26s} {B:3d} x {N:5d} -> {m:4d}: {us:8.1f} us per launch, {us / (m - 1):.3f} us per round  "
          f"(TPG_FPS_PRUNE={os.environ.get('TPG_FPS_PRUNE', '1')})", flush=True)
