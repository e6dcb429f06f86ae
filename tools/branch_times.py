"""Wall time of the graphed step with and without the discriminator updates (odd iterations
replay the generator-only graph): how long is the generator chain alone?  GPU box."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
argv_saved, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(b)
torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
models = b.build(dev, capturable=True)
clips = [b.fluid_clip(8, 4096, 8, 3, seed=s, device=dev) for s in range(4)]
from tpgan_amd.gan_step_graph import GraphedFluidStep
G, Ds, Dt, opts = models
step = GraphedFluidStep(G, Ds, Dt, opts, b.OPT, clips[0][0], clips[0][1], 1.0, torch.bfloat16, None)
which = argv_saved[1] if len(argv_saved) > 1 else "both"      # "odd": only the generator-only graph (for rocprofv3 --stats)
cases = ((13, "G only"),) if which == "odd" else ((12, "G + both D updates"), (13, "G only"))
for n_iter, label in cases:
    for i in range(3):
        step(*clips[i % 4], n_iter)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        step(*clips[i % 4], n_iter)
    torch.cuda.synchronize()
    print(f"{label:22s} {(time.perf_counter() - t0) / 20 * 1e3:7.2f} ms per step")
