"""The generator's forward as the ORDERED list of aten / custom ops that reach the device, each with the innermost
frame of this package that issued it (a TorchDispatchMode; torch.profiler's with_stack gives no Python frames on this
build).  Read next to tools/gen_forward_trace.py, which has the same launches with their durations.  GPU box.

    python tools/gen_forward_ops.py
"""
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

VIEWS = ("view", "reshape", "transpose", "permute", "slice", "select", "unsqueeze", "squeeze", "expand", "t.default",
         "detach", "alias", "unbind", "split", "as_strided", "empty", "_unsafe_view", "size", "stride", "is_", "sym_")
dev = torch.device("cuda", 0)
np.random.seed(0)
G, Ds, Dt, opts = configs.build_models("cfg2", dev, capturable=True)
low, high = configs.make_clip("cfg2", seed=0, device=dev)
stacked = torch.cat([low[1], low[0], low[2]], 0)


def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return G.body(stacked, stacked)


fwd()
seq = []


class Order(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(v in name for v in VIEWS):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "temporal-pointcloud" in fr.filename and "gen_forward_ops" not in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            shapes = [tuple(a.shape) if isinstance(a, torch.Tensor) else None for a in args[:2]]
            seq.append((name, where, shapes))
        return func(*args, **(kwargs or {}))


with Order():
    fwd()
torch.cuda.synchronize()
print(len(seq), "ops")
for name, where, shapes in seq:
    print(f"{name:32s} {where:46s} {shapes}")
