"""Where do the small aten launches of the generator-only body come from?  A TorchDispatchMode counts every
aten op of one eager run of the captured body by (op, dtypes, innermost frame inside this repo); forward and the
Python side of the backward (the autograd engine's C++ thread has no frames: those show as "<autograd>").  GPU box.

    python tools/copy_sources.py [op-substring ...]        # default: copy_ _to_copy cat leaky_relu add mul fill_
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

import tpgan_amd  # noqa: F401
from tpgan_amd import configs

want = sys.argv[1:] or ["copy_", "_to_copy", "cat", "leaky_relu", "add", "mul", "fill_", "sum", "clone", "zeros", "empty_like"]
torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
models = configs.build_models("cfg2", dev, capturable=True)
clips = [configs.make_clip("cfg2", seed=s, device=dev) for s in range(2)]
step = configs.graphed_step("cfg2", models, clips[0], amp_dtype=torch.bfloat16)
step._load(*clips[1])
step._run_eager(False)
torch.cuda.synchronize()
count = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if any(w in name for w in want):
            where = "<autograd>"
            for fr in reversed(traceback.extract_stack()):
                if "temporal-pointcloud" in fr.filename and "tools" not in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            sig = "" if t is None else f"{str(t.dtype)[6:]}{tuple(t.shape)}"
            if t is not None and t.is_cuda and t.numel() > 0:
                count[(name, where, sig)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step._run_eager(False)
torch.cuda.synchronize()
for (name, where, sig), n in count.most_common(70):
    print(f"{n:4d}  {name:28s} {where:48s} {sig}")
