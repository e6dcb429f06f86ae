"""The generator's forward is the head of the step's critical path (nothing but the real clouds'
index plans can run beside it).  This lists its kernels in launch order with durations and the
line of this package that launched them, and times the same forward as a hipGraph replay.  GPU box.

    python tools/gen_forward_trace.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

import tpgan_amd  # noqa: F401,E402
from tpgan_amd import configs  # noqa: E402

dev = torch.device("cuda", 0)
np.random.seed(0)
G, Ds, Dt, opts = configs.build_models("cfg2", dev, capturable=True)
low, high = configs.make_clip("cfg2", seed=0, device=dev)
stacked = torch.cat([low[1], low[0], low[2]], 0)


def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return G.body(stacked, stacked)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay()
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("G.body forward, %d clouds x %d points: %.3f ms per hipGraph replay" % (stacked.shape[0], stacked.shape[1], e0.elapsed_time(e1) / 20))

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fwd()
    torch.cuda.synchronize()


def own_frame(evt):
    p = evt
    while p is not None:
        for fr in (p.stack or []):
            if ("temporal-pointcloud" in fr or "tpgan_amd" in fr) and "site-packages" not in fr:
                return fr.replace(ROOT + "/", "").replace("temporal-pointcloud-upsampling-gan_amd/", "")
        p = p.cpu_parent
    return "?"


rows = []
for evt in prof.events():
    kernels = getattr(evt, "kernels", None)
    if not kernels or any(getattr(c, "kernels", None) for c in evt.cpu_children):
        continue
    for k in kernels:
        rows.append((evt.time_range.start, k.duration, k.name[:60], evt.name, own_frame(evt)))
rows.sort()
print("%d kernels, %.3f ms of kernel time" % (len(rows), sum(r[1] for r in rows) / 1e3))
for _, dur, kname, op, where in rows:
    print("%8.1f us  %-60s %-28s %s" % (dur, kname, op[:28], where))
