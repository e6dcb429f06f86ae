"""BASELINE.json's cfg4 and cfg5 through the HIP path ("configs never run by a test").

cfg3 is cfg2's per-rank workload on 8 ranks (tests/test_ddp_*); cfg2 itself is what every other
GPU test and bench.py run.  Here: one rank's shard of cfg5 (N_hi = 16384, T = 5, r = 4) and cfg4
(N_hi = 2048, T = 8, r = 4), eager and replayed from hipGraphs, at reduced batch so the pair of
tests stays within a minute of GPU time; the full-batch runs are `bench.py --config`."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _finite(d):
    return all(np.isfinite(v) for v in d.values())


@pytest.mark.parametrize("name,batch", [("cfg5shard", 2), ("cfg4", 4)])
def test_config_steps_eager_and_replayed(name, batch):
    from tpgan_amd import configs
    dev = torch.device("cuda", 0)
    A = configs.build_models(name, dev, seed=5, capturable=True)
    for m in list(A[1].modules()) + list(A[2].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0           # (mask parity of replay and eager is test_graph_gpu's subject; here: shapes)
    Bm = copy.deepcopy(A[:3])
    Bm = (*Bm, tuple(torch.optim.Adam(m.parameters(), lr=g.param_groups[0]["lr"], capturable=True)
                     for m, g in zip((Bm[0], Bm[2], Bm[1]), A[3])))
    clips = [configs.make_clip(name, batch=batch, seed=s, device=dev) for s in (1, 2)]
    spec = configs.SPECS[name]
    assert clips[0][1][0].shape == (batch, spec["points"], 3) and len(clips[0][1]) == spec["frames"]
    assert clips[0][0][0].shape == (batch, spec["points"] // spec["ratio"], 3)
    stepper = configs.graphed_step(name, Bm, clips[0], amp_dtype=torch.bfloat16)
    configs.seed_host_rng(3)
    le = configs.eager_step(name, A, clips[0], 12, amp_dtype=torch.bfloat16)
    configs.seed_host_rng(3)
    lg = stepper(clips[0][0], clips[0][1], 12)
    print(name, "eager :", le)
    print(name, "replay:", lg)
    assert _finite(le) and _finite(lg) and le["tempo_D_loss"] > 0 and lg["tempo_D_loss"] > 0
    assert set(le) == set(lg)
    # same state, same host draws, bf16 in both: the RNG-free Chamfer term agrees tightly
    k = "Chamfer_distance_no_norm"
    assert abs(le[k] - lg[k]) <= 2e-3 * max(1.0, abs(le[k])), (le[k], lg[k])
    # the GAN terms of an untrained step sit on discrete decisions that a 1e-7 change of the inputs flips
    # (tests/test_graph_gpu.py::test_step_sensitivity...): the eager step and the re-organised replayed body
    # agree on them only loosely
    for k in ("tempo_G_loss", "tempo_D_loss", "spatial_G_loss", "spatial_D_loss"):
        assert abs(le[k] - lg[k]) <= 0.5, (k, le[k], lg[k])
    for it, c in ((13, clips[1]), (14, clips[0])):
        lg = stepper(c[0], c[1], it)
        assert _finite(lg) and (lg["tempo_D_loss"] > 0) == (it % 2 == 0)


def test_cfg5_discriminators_forward_backward_at_16384_points():
    """Both discriminators on N_hi = 16384 clouds, fp32, gradients to the clouds and the weights
    (the inverted index of the first level has more destination rows than 64 KB of LDS
    counters): outputs of the default order against the reference order (grouping_operation route)."""
    from tpgan_amd.set_abstraction import FluidSpatialDis, FluidTempoDis, reference_order
    from tpgan_amd.synthetic import fluid_clip
    dev = torch.device("cuda", 0)
    _, high = fluid_clip(2, 16384, 4, 5, seed=3, device=dev)
    for make, run in ((FluidSpatialDis, lambda m, xs: m(xs[2])), (lambda: FluidTempoDis(5), lambda m, xs: m(xs, 0.10))):
        outs = []
        for ref in (True, False):
            torch.manual_seed(9)
            m = make().to(dev).train()                  # batch statistics (an untrained net's running ones: logits ~1e23)
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            xs = [h.clone().requires_grad_(True) for h in high]
            with reference_order(ref):
                y = run(m, xs)
            y.sum().backward()
            gx = [x.grad for x in xs if x.grad is not None]
            assert len(gx) >= 1 and all(torch.isfinite(g).all() for g in gx)
            outs.append((y.detach(), torch.cat([g.reshape(-1) for g in gx]),
                         torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])))
        (y0, gx0, gp0), (y1, gx1, gp1) = outs
        print("cfg5 discriminator, reference order vs default order: logits", y0.flatten().tolist(), y1.flatten().tolist())
        # train-mode logits of a 2-clip batch: the head's BatchNorm1d is a sign network (5e-3, as in
        # tests/test_golden_models.py); gradients in L2 (arg-max near-ties may flip between the orders)
        assert float((y0 - y1).abs().max()) <= 5e-3 * max(1.0, float(y0.abs().max()))
        for a, b in ((gx0, gx1), (gp0, gp1)):
            assert torch.isfinite(a).all() and torch.isfinite(b).all()
            rel = float((a - b).norm() / a.norm().clamp_min(1e-12))
            print("   gradient relative L2 difference", rel)
            assert rel <= 5e-2, rel


def test_rollout_on_a_large_scene_matches_the_reference_order():
    """SURVEY section 8 row f3 at the size it is about: `SRNet.forward_with_context` on ONE cloud of 20000 points
    (upsampling_network.py:159-174).  The feature-space searches run the matrix-core filter, the 3-D search the
    grid; the reference order of operations (grouping_operation route) must see the same neighbours, so the two
    forms agree to summation order, frame after frame (the running mask average included)."""
    from tpgan_amd.set_abstraction import reference_order
    from tpgan_amd.srnet import SRNet
    from tpgan_amd.synthetic import fluid_clip, force_all_keep
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    net = force_all_keep(SRNet(3, 128, upsample_ratio=8)).to(dev).eval()
    low, _ = fluid_clip(1, 20000, 1, 2, seed=8, device=dev)
    outs = []
    for ref in (False, True):
        hist, frames = [], []
        with torch.no_grad(), reference_order(ref):
            for x in low:
                out, hist = net.forward_with_context(x, x, hist)
                frames.append(out)
        outs.append(frames)
    for a, b in zip(*outs):
        assert a.shape == (1, 8 * 20000, 3) and torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), float((a - b).abs().max())
