"""Which source lines of the step issue which aten ops (forward side)?  GPU box.

A TorchDispatchMode counts every aten op that reaches the device by the innermost frame inside
this package; companion of tools/kernel_sources.py (which prices the kernels and names the
autograd nodes of the backward side).

    python tools/op_sources.py [op substring ...]     # default: copy_ _to_copy div add cat fill mm
"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import tpgan_amd  # noqa: F401,E402
from tpgan_amd import configs  # noqa: E402

argv = sys.argv
wanted = argv[1:] or ["copy_", "_to_copy", "div", "add", "cat", "fill", "mm", "zeros", "ones", "clone", "leaky"]

torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
models = configs.build_models("cfg2", dev, capturable=True)
clips = [configs.make_clip("cfg2", seed=s, device=dev) for s in range(2)]
step = configs.graphed_step("cfg2", models, clips[0], amp_dtype=torch.bfloat16)
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
