"""Which lines of the step issue which aten ops -- forward AND backward side?  GPU box.

A TorchDispatchMode counts every aten op that reaches the device.  Forward ops are attributed to the innermost
frame inside this package; backward ops to the autograd node that ran them AND to the package line that created
that node in the forward (anomaly mode records the forward stack in the node's metadata).  View-only ops are
skipped.  Companion of tools/op_sources.py (forward only) and tools/kernel_sources.py (profiler based).

    python tools/launch_inventory.py [gonly|full [config [N]]]      # N = rows of the table (default 140)
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
update_D = not (len(argv) > 1 and argv[1] == "gonly")
config = argv[2] if len(argv) > 2 else "cfg2"
rows = int(argv[3]) if len(argv) > 3 else 140

VIEWS = ("view", "reshape", "slice.Tensor", "select.int", "transpose", "aten.t.", "expand", "as_strided", "detach",
         "alias", "unsqueeze", "squeeze", "permute", "narrow", "split", "unbind", "empty", "sym_", "size", "stride",
         "is_", "_unsafe_view", "lift_fresh", "unfold", "record_stream", "_local_scalar", "chunk", "numel",
         "result_type", "can_cast", "_has_", "set_.", "resize_", "storage_offset", "dim.", "prim.", "item")

torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
models = configs.build_models(config, dev, capturable=True)
clips = [configs.make_clip(config, seed=s, device=dev) for s in range(2)]
step = configs.graphed_step(config, models, clips[0], amp_dtype=torch.bfloat16)
step._load(*clips[1])
step._run_eager(update_D)
torch.cuda.synchronize()

count = collections.Counter()


def pkg_frame(frames):
    for fr in reversed(frames):
        fn = fr.filename if hasattr(fr, "filename") else fr
        if ("temporal-pointcloud" in fn or "torch/optim" in fn) and "launch_inventory" not in fn:
            if hasattr(fr, "filename"):
                return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
            return fn
    return None


def fwd_site(node):
    tb = node.metadata.get("traceback_") if node is not None else None
    if not tb:
        return None
    for line in reversed(tb):                     # '  File "...", line N, in f\n    code'
        head = line.strip().split("\n")[0]
        if "temporal-pointcloud" in head:
            try:
                path, rest = head.split('"')[1], head.split('"')[2]
                ln = rest.split(",")[1].strip().split(" ")[1]
                fn = rest.split(" in ")[-1]
                return "%s:%s %s" % (os.path.basename(path), ln, fn)
            except Exception:
                return head
    return None


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            node = torch._C._current_autograd_node()
            here = pkg_frame(traceback.extract_stack()[:-1])
            if node is not None:
                where = "bwd %-28s <- %s" % (node.name().replace("torch::autograd::", "")[:28],
                                             here if (here and "backward" in here) else (fwd_site(node) or here or "?"))
            else:
                where = "fwd " + (here or "?")
            count[(name.replace("aten.", ""), where)] += 1
        return func(*args, **(kwargs or {}))


with torch.autograd.set_detect_anomaly(True, check_nan=False):
    with Count():
        step._run_eager(update_D)
torch.cuda.synchronize()
print(f"{sum(count.values())} device ops in one eager body ({'full' if update_D else 'generator only'}, {config})")
for (name, where), n in count.most_common(rows):
    print(f"{n:5d}  {name:30s} {where}")
