"""Wall time of the graphed step with and without the discriminator updates (odd iterations replay the
generator-only graph), with the fused MFMA tails on and off, in ONE process on one box (interleaved
rounds: box-to-box spread is larger than the differences looked at).  GPU box.

    python tools/branch_times.py [config]         # cfg2 (default), cfg4, cfg5shard
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import configs, graph_conv, set_abstraction

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
clips = [configs.make_clip(name, seed=s, device=dev) for s in range(4)]
steppers = {}
for fused in (True, False):
    set_abstraction.FUSED_TAILS[0] = graph_conv.FUSED_EDGE_TAILS[0] = fused
    models = configs.build_models(name, dev, capturable=True)
    steppers[fused] = configs.graphed_step(name, models, clips[0], amp_dtype=torch.bfloat16)
set_abstraction.FUSED_TAILS[0] = graph_conv.FUSED_EDGE_TAILS[0] = True
res = {}
for rnd in range(3):
    for fused in (True, False):
        for n_iter, label in ((12, "G + both D updates"), (13, "G only")):
            st = steppers[fused]
            for i in range(2):
                st(*clips[i % 4], n_iter)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                st(*clips[i % 4], n_iter)
            torch.cuda.synchronize()
            res.setdefault((fused, label), []).append((time.perf_counter() - t0) / 20 * 1e3)
for (fused, label), v in res.items():
    print(f"{name} fused tails {'on ' if fused else 'off'} {label:20s} " + "  ".join(f"{x:6.2f}" for x in v) + "  ms per step")
