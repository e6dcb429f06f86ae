"""Which source lines of the step issue which aten ops (forward side)?  GPU box.

A TorchDispatchMode counts every aten op that reaches the device by the innermost frame inside
this package; companion of tools/kernel_sources.py (which prices the kernels and names the
autograd nodes of the backward side).

    python tools/op_sources.py [op substring ...]     # default: copy_ _to_copy div add cat fill mm
"""
import collections
import importlib.util
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(b)
wanted = argv[1:] or ["copy_", "_to_copy", "div", "add", "cat", "fill", "mm", "zeros", "ones", "clone", "leaky"]

torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
models = b.build(dev, capturable=True)
clips = [b.fluid_clip(8, 4096, 8, 3, seed=s, device=dev) for s in range(2)]
from tpgan_amd.gan_step_graph import GraphedFluidStep
G, Ds, Dt, opts = models
step = GraphedFluidStep(G, Ds, Dt, opts, b.OPT, clips[0][0], clips[0][1], 1.0, torch.bfloat16, None)
step._load(*clips[1])
step._run_eager(True)
torch.cuda.synchronize()

count = collections.Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(w in name for w in wanted):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ("temporal-pointcloud" in fr.filename or "torch/optim" in fr.filename) and "op_sources" not in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            count[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Count():
    step._run_eager(True)
torch.cuda.synchronize()
for (name, where), n in count.most_common(90):
    print(f"{n:5d}  {name:34s} {where}")
