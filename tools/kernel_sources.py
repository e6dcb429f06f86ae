"""Which lines of the step launch the SMALL kernels?  (torch.profiler, eager step, GPU box)

Every aten op that launches a kernel is attributed to the innermost frame inside this repo
(forward) or to the autograd node that ran it (backward), and counted.  Output: one table,
sorted by launches per step -- the input for deciding what to fuse next.

    python tools/kernel_sources.py [max_us [gonly|full [config]]]      # only kernels shorter than max_us (default 8)
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

import tpgan_amd  # noqa: F401,E402
from tpgan_amd import configs  # noqa: E402

argv = sys.argv
max_us = float(argv[1]) if len(argv) > 1 else 8.0
update_D = not (len(argv) > 2 and argv[2] == "gonly")      # "gonly": the generator-only body (the step's critical chain)
config = argv[3] if len(argv) > 3 else "cfg2"

torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
models = configs.build_models(config, dev, capturable=True)
clips = [configs.make_clip(config, seed=s, device=dev) for s in range(2)]
step = configs.graphed_step(config, models, clips[0], amp_dtype=torch.bfloat16)
step._load(*clips[1])
step._run_eager(update_D)                                   # the body the graph captured, run eagerly
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step._run_eager(update_D)
    torch.cuda.synchronize()


def own_frame(evt):
    p = evt
    while p is not None:                       # the Python stack hangs on the outermost op
        for fr in (p.stack or []):
            if ("temporal-pointcloud" in fr or "tpgan_amd" in fr) and "site-packages" not in fr:
                return fr.replace(ROOT + "/", "").replace("temporal-pointcloud-upsampling-gan_amd/", "")
        p = p.cpu_parent
    return None


def autograd_node(evt):
    p = evt
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function: "):
            return "bwd " + p.name.split(": ", 1)[1]
        p = p.cpu_parent
    return None


KERNEL_FILTER = [f for f in os.environ.get("KS_KERNEL", "").split(",") if f]     # only kernels whose name has one of these
count = collections.Counter()
time_us = collections.Counter()
for evt in prof.events():
    kernels = getattr(evt, "kernels", None)
    if not kernels:
        continue
    # leaf-most op only: skip if a child op also owns kernels
    if any(getattr(c, "kernels", None) for c in evt.cpu_children):
        continue
    for k in kernels:
        dur = k.duration
        if dur > max_us:
            continue
        if KERNEL_FILTER and not any(f in k.name for f in KERNEL_FILTER):
            continue
        where = autograd_node(evt) or own_frame(evt) or "?"
        key = (evt.name, where)
        count[key] += 1
        time_us[key] += dur
total = sum(count.values())
print(f"{total} kernels shorter than {max_us} us in one eager step, {sum(time_us.values()) / 1e3:.2f} ms")
for key, n in count.most_common(70):
    print(f"{n:5d}  {time_us[key] / 1e3:6.3f} ms  {key[0]:28s} {key[1]}")
if os.environ.get("KS_DUMP_STACK"):
    for evt in prof.events():
        if getattr(evt, "kernels", None) and evt.stack:
            print(evt.name, evt.stack[:12]); break
