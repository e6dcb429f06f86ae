"""What would the step cost if the REAL-cloud half of both discriminator updates ran somewhere else (= at the
head of the step, beside the generator's forward)?  Times the replayed cfg2 step with the updates reduced to
their fake batch (measurement aid only: the result of such a step is not the training step's).  GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import configs
from tpgan_amd.set_abstraction import _SpatialDis, _TempoDis

torch.backends.cudnn.enabled = False
dev = torch.device("cuda", 0)
np.random.seed(0)
clips = [configs.make_clip("cfg2", seed=s, device=dev) for s in range(4)]


def timed(st, n_iter=12):
    for i in range(3):
        st(*clips[i % 4], n_iter)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        st(*clips[i % 4], n_iter)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


full = configs.graphed_step("cfg2", configs.build_models("cfg2", dev, capturable=True), clips[0], amp_dtype=torch.bfloat16)
# fake-only variant: forward_passes of an update sees [fake, real] -> run the fake pass only, return its logits twice
orig_t, orig_s = _TempoDis.forward_passes, _SpatialDis.forward_passes


def fp_t(self, pos_lsts, cutoff, plan=None):
    if len(pos_lsts) == 2:
        out = orig_t(self, pos_lsts[:1], cutoff, plan=None)
        return [out[0], out[0].detach()]
    return orig_t(self, pos_lsts, cutoff, plan=plan)


def fp_s(self, pos_list, plan=None):
    if len(pos_list) == 2:
        out = orig_s(self, pos_list[:1], plan=None)
        return [out[0], out[0].detach()]
    return orig_s(self, pos_list, plan=plan)


_TempoDis.forward_passes, _SpatialDis.forward_passes = fp_t, fp_s
half = configs.graphed_step("cfg2", configs.build_models("cfg2", dev, capturable=True), clips[0], amp_dtype=torch.bfloat16)
_TempoDis.forward_passes, _SpatialDis.forward_passes = orig_t, orig_s
for rnd in range(2):
    print(f"full step {timed(full):6.2f} ms   updates on the fake batch only {timed(half):6.2f} ms   generator only {timed(full, 13):6.2f} ms")
