"""Inference rollout (`SRNet.forward_with_context`, upsampling_network.py:159-174) on LARGE clouds: SURVEY
section 8 row f3 is about 1e4..1e5-point scenes, the training clips stop at 2048 low-resolution points.
One cloud, no gradients, bf16 autocast and fp32; per frame time, the generated point count, and the share
of the three searches (first EdgeConv's 3-D kNN, the two IDGCN blocks' 32-dim feature-space kNN).  GPU box.

    python tools/rollout_time.py [points ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import ops
from tpgan_amd.srnet import SRNet
from tpgan_amd.synthetic import fluid_clip, force_all_keep

sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536, 100000]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = force_all_keep(SRNet(3, 128, upsample_ratio=8)).to(dev).eval()

for n in sizes:
    low, _ = fluid_clip(1, n, 1, 4, seed=n, device=dev)          # four frames of an n-point low-resolution scene
    for dtype in (torch.bfloat16, None):
        hist = []
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
            out, hist = net.forward_with_context(low[0], low[0], hist)          # warm-up (workspaces, GEMM plans)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for x in low[1:]:
                out, hist = net.forward_with_context(x, x, hist)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (len(low) - 1) * 1e3
            timer = ops.OpTimer()
            prev = ops.set_timer(timer)
            net.forward_with_context(low[0], low[0], [])
            ops.set_timer(prev)
            torch.cuda.synchronize()
        knn = sum(v["total_ms"] for k, v in timer.summary().items() if k.startswith("knn") or k.startswith("frnn"))
        print(f"rollout {n:7d} points {'bf16' if dtype else 'fp32'}: {ms:8.2f} ms per frame -> {out.shape[1]} points "
              f"({n / ms * 1e3 / 1e6:.2f} M input points/s); neighbour searches {knn:.2f} ms; "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
