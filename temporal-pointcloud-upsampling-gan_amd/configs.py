"""The BASELINE.json workloads as (networks, optimizers, synthetic clips, step) bundles.

One place for what bench.py times and what tests/test_configs_gpu.py runs through the HIP path:

  cfg2       4096-pt x3-frame fluid clips, batch 8 per GPU, full G+D step (the headline metric;
             cfg3 is the same per-rank workload on 8 ranks)
             SRNet(3,128, r=8) + FluidTempoDis(3) + FluidSpatialDis       train_fluid/train_tempo.py
  cfg4       MSR-Action3D-like 2048-pt x8-frame clips, batch 8, ratio 4 (N_lo = 512)
             NoMaskSRNet(3,128, r=4) + ActionTempoDis(8) + ActionSpatialDis   train_action/train_msr.py:98
             (the reference never interpolates in time: "4x temporal upsample" is read as r = 4 with an
             8-frame temporal discriminator)
  cfg5shard  ONE rank's share of cfg5: 16384-pt x5-frame dense fluid clips, r = 4 (N_lo = 4096),
             SRNet(3,128, r=4) + FluidTempoDis(5) + FluidSpatialDis; BASELINE quotes batch 8 per GPU on
             8 GPUs -- `batch` clips here
"""
from argparse import Namespace

import numpy as np
import torch

from .set_abstraction import ActionSpatialDis, ActionTempoDis, FluidSpatialDis, FluidTempoDis
from .srnet import NoMaskSRNet, SRNet
from .synthetic import action_clip, fluid_clip, force_all_keep

FLUID_OPT = Namespace(use_vel=False, in_node_feats=3, cutoff=0.025, R=0.10, w=0.5)
ACTION_OPT = Namespace(R=2.0, w=2.0)

SPECS = {
    "cfg2": dict(kind="fluid", points=4096, ratio=8, frames=3, batch=8,
                 label="cfg2: 4096-pt x3-frame fluid clips, full G+D adversarial step (SRNet(3,128) + "
                       "FluidTempoDis(3) + FluidSpatialDis, Adam), all-keep mask regime, gate open, even iteration"),
    "cfg4": dict(kind="action", points=2048, ratio=4, frames=8, batch=8,
                 label="cfg4: MSR-Action3D-like 2048-pt x8-frame clips, ratio 4, full G+D step (NoMaskSRNet(3,128,r=4) + "
                       "ActionTempoDis(8) + ActionSpatialDis, Adam), even iteration"),
    "cfg5shard": dict(kind="fluid", points=16384, ratio=4, frames=5, batch=8,
                      label="cfg5 (one rank's shard): 16384-pt x5-frame dense fluid clips, 4x spatial upsample, full "
                            "G+D step (SRNet(3,128,r=4) + FluidTempoDis(5) + FluidSpatialDis, Adam), all-keep mask "
                            "regime, gate open, even iteration"),
}


def build_models(name, device, seed=1, capturable=False, lr=3e-4):
    """-> (G, Ds, Dt, (opt_G, opt_Dt, opt_Ds)) of workload `name`, random-init (no checkpoints ship)."""
    spec = SPECS[name]
    torch.manual_seed(seed)
    if spec["kind"] == "fluid":
        G = force_all_keep(SRNet(3, 128, upsample_ratio=spec["ratio"])).to(device)
        Ds, Dt = FluidSpatialDis().to(device), FluidTempoDis(spec["frames"]).to(device)
    else:
        G = NoMaskSRNet(3, 128, upsample_ratio=spec["ratio"]).to(device)
        Ds, Dt = ActionSpatialDis().to(device), ActionTempoDis(spec["frames"]).to(device)
    # capturable for the hipGraph path; fused = torch's single-launch multi-tensor Adam (the foreach
    # form divides by per-parameter 0-dim step tensors one launch per parameter)
    kw = {"capturable": True, "fused": torch.device(device).type == "cuda"} if capturable else {}

    def adam(params, lr_):
        try:
            return torch.optim.Adam(params, lr=lr_, **kw)
        except (RuntimeError, ValueError):
            return torch.optim.Adam(params, lr=lr_, **{k: v for k, v in kw.items() if k != "fused"})
    opts = (adam(list(G.parameters()), lr), adam(list(Dt.parameters()), 0.33 * lr), adam(list(Ds.parameters()), 0.33 * lr))
    return G, Ds, Dt, opts


def make_clip(name, batch=None, points=None, seed=1234, device="cpu"):
    """One synthetic clip (lowres_pos_lst, highres_pos_lst) of workload `name`."""
    spec = SPECS[name]
    batch = spec["batch"] if batch is None else batch
    points = spec["points"] if points is None else points
    if spec["kind"] == "fluid":
        return fluid_clip(batch, points, spec["ratio"], spec["frames"], seed=seed, device=device)
    return action_clip(batch, points, spec["ratio"], spec["frames"], seed=seed, device=device)


def opt_of(name):
    return FLUID_OPT if SPECS[name]["kind"] == "fluid" else ACTION_OPT


def eager_step(name, models, clip, n_iter=12, sync=None, amp_dtype=None):
    """One eager step of workload `name` (gate forced open for the fluid workloads: benchmark
    regime)."""
    from .gan_step import tempo_gan_step, tempo_gan_step_no_mask
    G, Ds, Dt, (og, ot, os_) = models
    low, high = clip
    if SPECS[name]["kind"] == "fluid":
        return tempo_gan_step(G, Ds, Dt, low, None, high, None, 1.0, FLUID_OPT, n_iter, og, ot, os_,
                              sync=sync, amp_dtype=amp_dtype, force_gate=True)
    return tempo_gan_step_no_mask(G, Ds, Dt, low, high, ACTION_OPT, n_iter, og, ot, os_, sync=sync,
                                  amp_dtype=amp_dtype)


def graphed_step(name, models, clip, amp_dtype=None, sync=None):
    """The hipGraph-replayed step of workload `name`, captured on `clip`'s shapes."""
    from .gan_step_graph import GraphedActionStep, GraphedFluidStep
    G, Ds, Dt, opts = models
    cls = GraphedFluidStep if SPECS[name]["kind"] == "fluid" else GraphedActionStep
    return cls(G, Ds, Dt, opts, opt_of(name), clip[0], clip[1], 1.0, amp_dtype, sync)


def seed_host_rng(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)
