"""Action-clip step (tempo_gan_step_no_mask, BASELINE cfg4 family): eager vs hipGraph replay.  GPU box.

    python tools/action_step_time.py [T] [ratio]     # default T = 3 frames, ratio 16 (the reference's MSR setting)
"""
import os
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd.gan_step import tempo_gan_step_no_mask
from tpgan_amd.gan_step_graph import GraphedActionStep
from tpgan_amd.set_abstraction import ActionSpatialDis, ActionTempoDis
from tpgan_amd.srnet import NoMaskSRNet
from tpgan_amd.synthetic import action_clip

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ratio = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
G = NoMaskSRNet(3, 128, upsample_ratio=ratio).to(dev)
Ds, Dt = ActionSpatialDis().to(dev), ActionTempoDis(T).to(dev)
opt = Namespace(R=2.0, w=2.0)
clips = [action_clip(8, 2048, ratio, T, seed=s, device=dev) for s in range(4)]
kw = dict(lr=1e-4, capturable=True, fused=True)
opts = (torch.optim.Adam(G.parameters(), **kw), torch.optim.Adam(Dt.parameters(), **kw), torch.optim.Adam(Ds.parameters(), **kw))
for mode in ("eager", "graph"):
    if mode == "graph":
        stepper = GraphedActionStep(G, Ds, Dt, opts, opt, clips[0][0], clips[0][1], 1.0, torch.bfloat16, None)
        run = lambda c, it: stepper(c[0], c[1], it)          # noqa: E731
    else:
        run = lambda c, it: tempo_gan_step_no_mask(G, Ds, Dt, c[0], c[1], opt, it, *opts, amp_dtype=torch.bfloat16)  # noqa: E731
    for i in range(3):
        out = run(clips[i % 4], 12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        out = run(clips[i % 4], 12)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"action step B=8 N_hi=2048 T={T} r={ratio} bf16  {mode:6s} {dt * 1e3:7.2f} ms/step = {1 / dt:6.1f} steps/s   "
          f"{ {k: round(v, 3) for k, v in out.items()} }")
