"""Replay only the generator-only graph (odd iterations) N times: for rocprofv3 --kernel-trace --stats, to see
what the step's critical chain (generator forward, losses, the two discriminator forwards of the generator
step, their data gradients, the generator's backward, Adam) is made of.  GPU box.

    rocprofv3 --kernel-trace --stats -d out -- python3 tools/gonly_profile.py [config] [replays]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import configs

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
clips = [configs.make_clip(name, seed=s, device=dev) for s in range(2)]
models = configs.build_models(name, dev, capturable=True)
st = configs.graphed_step(name, models, clips[0], amp_dtype=torch.bfloat16)
for i in range(2):
    st(*clips[i % 2], 13)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    st(*clips[i % 2], 13)
torch.cuda.synchronize()
print(f"{name}: generator-only replay {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step (under the profiler if one is attached)")
