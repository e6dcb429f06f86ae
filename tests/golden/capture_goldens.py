"""Generates tests/golden/*.npz by running the REFERENCE's own Python in the build container.

    python tests/golden/capture_goldens.py            # writes the fixtures next to this file

What runs: `/root/reference` (read-only, imported unmodified; never copied, never shipped)
with `pointnet2_ops`, `pytorch3d`, `frnn`, `chamferdist` resolved to this repo's
import-compatible modules, whose CPU tensors are served by the oracle (oracle/tpgref.c).
`dgl`, `numba`, `open3d`, `emd` are inert import shims (tests/golden/_import_shims).
`Tensor.cuda()` / `Module.cuda()` are patched to identity because the reference hard-codes
`.cuda()` (loss.py:174, train_step_final.py:30,156-157).

What is stored: inputs and expected outputs only (arrays), plus per-tensor checksums of the
seeded random weights so the tests can prove that the build's modules, constructed under
the same seed, hold bit-identical parameters under identical names.  The reference has no
pretrained checkpoints (.MISSING_LARGE_BLOBS) and no fixtures of its own.

This script cannot run on the GPU box (there is no /root/reference there); the tests only
read the committed .npz files.
"""
import os
import sys
import warnings
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "_import_shims"))

import tpgan_amd  # noqa: E402

tpgan_amd.install_compat()
sys.path.insert(2, REFERENCE)
from oracle import torch_backend  # noqa: E402

torch_backend.install()
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
warnings.simplefilter("ignore")

import discriminator as ref_dis  # noqa: E402
import loss as ref_loss  # noqa: E402
import train_step_final as ref_step  # noqa: E402
import upsampling_network as ref_net  # noqa: E402

from tpgan_amd.synthetic import action_clip, fluid_clip  # noqa: E402

torch.set_num_threads(8)


def checksums(module):
    """name -> (sum, abs-sum) in float64 for every state-dict entry."""
    out = {}
    for k, v in module.state_dict().items():
        v = v.detach().double()
        out[k] = np.array([v.sum().item(), v.abs().sum().item(), float(v.numel())])
    return out


def pack(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


def n(t):
    return t.detach().cpu().numpy()


def set_mask_head(net, mode, seed):
    """all-keep: mask == 1.  mixed: mask values straddle epsilon = 0.01 (pads with 999)."""
    last = net.filter_block.decoder[1]
    with torch.no_grad():
        if mode == "keep":
            last.weight.zero_()
            last.bias.fill_(1.0)
        elif mode == "mixed":
            g = torch.Generator().manual_seed(seed)
            last.weight.copy_(30.0 * torch.randn(last.weight.shape, generator=g))
            last.bias.fill_(-0.01)


# ----------------------------------------------------------------------------- generator
def capture_generator():
    out = {}
    torch.manual_seed(11)
    net = ref_net.SRNet(3, 128)
    out.update(pack("w", checksums(net)))
    low, _ = fluid_clip(2, 1024, 8, 1, seed=5)
    x = low[0]
    out["x"] = n(x)
    for mode in ("init", "keep", "mixed"):
        set_mask_head(net, mode, 3)
        for hard in (False, True):
            pos, mask, padded = net(x, x, hard_masking=hard)
            tag = f"{mode}/{'hard' if hard else 'soft'}"
            out[f"{tag}/pos"] = n(pos)
            out[f"{tag}/mask"] = n(mask)
            if padded is not None:
                out[f"{tag}/padded"] = n(padded)
    # six input features: kNN of the first layer runs on pos (upsampling_network.py:177-180)
    torch.manual_seed(12)
    net6 = ref_net.SRNet(6, 128)
    out.update(pack("w6", checksums(net6)))
    set_mask_head(net6, "keep", 0)
    f6 = torch.cat([x, 0.1 * torch.randn(x.shape, generator=torch.Generator().manual_seed(2))], dim=2)
    out["f6"] = n(f6)
    pos, mask, _ = net6(f6, x, hard_masking=False)
    out["six/pos"], out["six/mask"] = n(pos), n(mask)
    # rollout with context (upsampling_network.py:159-174)
    set_mask_head(net, "keep", 0)
    hist = []
    p1, hist = net.forward_with_context(x[:1], x[:1], hist)
    p2, hist = net.forward_with_context(x[1:], x[1:], hist)
    out["ctx/p1"], out["ctx/p2"] = n(p1), n(p2)
    save("generator_srnet", **out)

    out = {}
    torch.manual_seed(13)
    net = ref_net.NoMaskSRNet(3, 128, upsample_ratio=16)
    out.update(pack("w", checksums(net)))
    low, _ = action_clip(2, 2048, 16, 1, seed=6)
    out["x"] = n(low[0])
    pos, edge = net(low[0], low[0])
    out["pos"], out["edge"] = n(pos), n(edge)
    save("generator_nomask", **out)


# ------------------------------------------------------------------------- discriminators
def capture_discriminators():
    _, high = fluid_clip(2, 1024, 8, 3, seed=7)
    _, ahigh = action_clip(2, 1024, 16, 3, seed=8)
    out = {"fluid": np.stack([n(h) for h in high]), "action": np.stack([n(h) for h in ahigh])}
    specs = [("fluid_spatial", lambda: ref_dis.FluidSpatialDis(), lambda m: m(high[1])),
             ("fluid_tempo", lambda: ref_dis.FluidTempoDis(3), lambda m: m(list(high), 0.10)),
             ("action_spatial", lambda: ref_dis.ActionSpatialDis(), lambda m: m(ahigh[1])),
             ("action_tempo", lambda: ref_dis.ActionTempoDis(3), lambda m: m(list(ahigh), 2.0))]
    for i, (name, make, run) in enumerate(specs):
        torch.manual_seed(20 + i)
        m = make()
        out.update(pack(f"{name}/w", checksums(m)))
        m.train()
        torch.manual_seed(100 + i)          # dropout draws
        out[f"{name}/train"] = n(run(m))
        out.update(pack(f"{name}/w_after", checksums(m)))   # BN running stats, spectral-norm u/v
        m.eval()
        out[f"{name}/eval"] = n(run(m))
    # padded clouds: FPS hits on 999-dummies are replaced (discriminator.py:115-130)
    padded = high[1].clone()
    padded[0, 700:] = 999
    padded[1, 900:] = 999
    out["padded"] = n(padded)
    torch.manual_seed(24)
    m = ref_dis.FluidSpatialDis()
    m.eval()
    np.random.seed(77)
    out["fluid_spatial/padded_eval"] = n(m(padded))
    # T = 5 frames: 10 FlowEmbedding calls
    _, high5 = fluid_clip(1, 1024, 8, 5, seed=9)
    out["fluid5"] = np.stack([n(h) for h in high5])
    torch.manual_seed(25)
    m = ref_dis.FluidTempoDis(5)
    out.update(pack("fluid_tempo5/w", checksums(m)))
    m.eval()
    out["fluid_tempo5/eval"] = n(m(list(high5), 0.10))
    # ball_query_wrapper with a radius small enough to leave -1 slots before the kNN fill
    idx = ref_dis.ball_query_wrapper(0.04, 32, high[0][:, :256], high[1])
    out["bqw/idx"] = n(idx)
    save("discriminators", **out)


# ------------------------------------------------------------------------------- losses
def capture_losses():
    low, high = fluid_clip(2, 1024, 8, 1, seed=10)
    g = torch.Generator().manual_seed(4)
    pred = (high[0] + 0.01 * torch.randn(high[0].shape, generator=g)).requires_grad_(True)
    mask = torch.rand(2, 128, 1, generator=g).requires_grad_(True)
    out = {"low": n(low[0]), "high": n(high[0]), "pred": n(pred), "mask": n(mask)}
    for n_iter in (5, 12):
        total, cd, ml = ref_loss.tpugan_sr_loss(100., high[0], pred, low[0], mask, 0.025, n_iter)
        gp, gm = torch.autograd.grad(total, [pred, mask], allow_unused=True)
        out[f"it{n_iter}/total"], out[f"it{n_iter}/cd"], out[f"it{n_iter}/ml"] = n(total), n(cd), n(ml)
        out[f"it{n_iter}/grad_pred"] = n(gp)
        if gm is not None:
            out[f"it{n_iter}/grad_mask"] = n(gm)
    out["cd_unbatched"] = n(ref_loss.chamfer_distance_loss(high[0][0], pred[0]))
    save("losses", **out)


# --------------------------------------------------------------------------- train steps
# Stability probe: relative coordinate jitter and the loss change a seed may show under it.
# 1e-7 = fp32 rounding: what another summation order (another BLAS, a GPU) does to the inputs of
# the discrete decisions.  (At 3e-5 -- the size of what this repo's default order, first layer
# before the gather, does to the generator's output: W f_j - W f_i against W (f_j - f_i) rounds
# relative to |f|, not to |f_j - f_i| -- NO seed of 12 was stable: every clip moved some GAN loss by
# 6-20 %.  The GAN terms of an untrained step are chaotic at that level, which is why the default
# order is pinned piecewise -- generator, discriminators, losses, gradients of both orders -- and
# only the reference order is pinned on whole steps.)
JITTER = 1e-7
STABLE = 1e-4
STEP_BATCH = 4    # clips per step fixture: with 2 the heads' BatchNorm1d sees xhat = +-1 (a sign network)


def _ref_step(kind, clip_seed, noise_seed=None):
    """One step of the reference from seeded state on the seeded clip; noise_seed: the clip's
    coordinates multiplied by (1 + JITTER * N(0,1)) first (stability probe, see step_fixture)."""
    if kind == "action":
        torch.manual_seed(40)
        G = ref_net.NoMaskSRNet(3, 128, upsample_ratio=16)
        torch.manual_seed(41)
        Ds = ref_dis.ActionSpatialDis()
        torch.manual_seed(42)
        Dt = ref_dis.ActionTempoDis(3)
        low, high = action_clip(STEP_BATCH, 2048, 16, 3, seed=clip_seed)
        opt = Namespace(R=2.0, w=2.0)
    else:
        torch.manual_seed(30)
        G = ref_net.SRNet(3, 128)
        torch.manual_seed(31)
        Ds = ref_dis.FluidSpatialDis()
        torch.manual_seed(32)
        Dt = ref_dis.FluidTempoDis(3)
        if kind == "fluid_keep":
            set_mask_head(G, "keep", 0)
        low, high = fluid_clip(STEP_BATCH, 1024, 8, 3, seed=clip_seed)
        opt = Namespace(use_vel=False, in_node_feats=3, cutoff=0.025, R=0.10, w=0.5)
    if noise_seed is not None:
        g = torch.Generator().manual_seed(noise_seed)
        low = [x * (1 + JITTER * torch.randn(x.shape, generator=g)) for x in low]
        high = [x * (1 + JITTER * torch.randn(x.shape, generator=g)) for x in high]
    before = {tag: checksums(m) for tag, m in (("G", G), ("Ds", Ds), ("Dt", Dt))}
    # plain SGD so that parameter deltas are proportional to the gradients being compared
    og = torch.optim.SGD(G.parameters(), lr=0.05)
    ot = torch.optim.SGD(Dt.parameters(), lr=0.05)
    os_ = torch.optim.SGD(Ds.parameters(), lr=0.05)
    torch.manual_seed(500)
    np.random.seed(500)
    if kind == "action":
        losses = ref_step.tempo_gan_step_no_mask(G, Ds, Dt, list(low), list(high), opt, 12, og, ot, os_)
    else:
        losses = ref_step.tempo_gan_step(G, Ds, Dt, list(low), None, list(high), None, 1.0, opt, 12,
                                         og, ot, os_)
    return losses, before, (G, Ds, Dt), low, high


def step_fixture(kind, first_seed=31, tries=24):
    """A full adversarial step of the reference: losses + parameters after one SGD step.

    The GAN terms of a step of UNTRAINED networks can sit on a discrete decision (an FPS pick, a
    ball-query membership, a max-pool winner on the generated clouds) that a rounding-level change
    of the coordinates flips, moving a loss by 1e-2: measured on the first fixture of this repo
    (two clips, seed 31), whose `tempo_G_loss` read 0.96814 or 0.94807 under a 1e-7 jitter of the
    inputs -- and with two clips the heads' BatchNorm1d is a sign network on top of that.  Such a
    clip pins nothing but the reference's own summation order.  So the fixtures hold four clips
    and the clip seed is SELECTED: the first one for which six probe runs of the reference -- three
    with coordinates jittered by JITTER, one with oneDNN off, two with other thread counts --
    reproduce every loss of the clean run to STABLE (the steadiest of `tries` seeds otherwise; the
    measured sensitivity is stored as `jitter_sensitivity`).  Any
    faithful implementation (other GEMM order, other device, first layer before the gather) then
    lands on the same decisions."""
    best = None
    for clip_seed in range(first_seed, first_seed + tries):
        losses = _ref_step(kind, clip_seed)[0]
        worst = 0.0
        # probes: input jitter, and the reference's own arithmetic in another summation order (oneDNN
        # off = native convolutions; one thread = other GEMM blocking) -- the first batch-4 fixture
        # (fluid, seed 37) passed three jitters and still read 1.1048 / 1.6758 with 8 oneDNN threads
        # and 1.1117 / 1.7294 everywhere else (1 thread, oneDNN off, this repo on CPU and GPU)
        probes = [("jitter", 1), ("jitter", 2), ("jitter", 3), ("native", None), ("threads", 1), ("threads", 3)]
        for what, arg in probes:
            if what == "jitter":
                jl = _ref_step(kind, clip_seed, arg)[0]
            elif what == "native":
                torch.backends.mkldnn.enabled = False
                try:
                    jl = _ref_step(kind, clip_seed)[0]
                finally:
                    torch.backends.mkldnn.enabled = True
            else:
                torch.set_num_threads(arg)
                try:
                    jl = _ref_step(kind, clip_seed)[0]
                finally:
                    torch.set_num_threads(8)
            worst = max(worst, max(abs(jl[k] - losses[k]) / max(1.0, abs(losses[k])) for k in losses))
            if worst > STABLE:
                break
        print(f"{kind}: clip seed {clip_seed}: worst loss change under the probes {worst:.2e}", flush=True)
        if best is None or worst < best[0]:
            best = (worst, clip_seed)
        if worst <= STABLE:
            break
    worst, clip_seed = best
    losses, before, nets, low, high = _ref_step(kind, clip_seed)
    out = {"clip_seed": np.int64(clip_seed), "jitter_sensitivity": np.float64(worst)}
    for tag in ("G", "Ds", "Dt"):
        out.update(pack(f"w/{tag}", before[tag]))
    out["low"] = np.stack([n(x) for x in low])
    out["high"] = np.stack([n(x) for x in high])
    for k, v in losses.items():
        out[f"loss/{k}"] = np.float64(v)
    for tag, m in zip(("G", "Ds", "Dt"), nets):
        out.update(pack(f"w_after/{tag}", checksums(m)))
    print(kind, losses)
    save(f"step_{kind}", **out)


def capture_discriminators_t8():
    """cfg4's temporal discriminator: ActionTempoDis(8) over eight frames = 28 FlowEmbedding calls
    at seven depths (discriminator.py:286-322,325-402), train-mode logits + state, eval logits."""
    _, ahigh = action_clip(2, 1024, 16, 8, seed=50)
    out = {"action8": np.stack([n(h) for h in ahigh])}
    torch.manual_seed(60)
    m = ref_dis.ActionTempoDis(8)
    out.update(pack("action_tempo8/w", checksums(m)))
    m.train()
    torch.manual_seed(160)
    out["action_tempo8/train"] = n(m(list(ahigh), 2.0))
    out.update(pack("action_tempo8/w_after", checksums(m)))
    m.eval()
    out["action_tempo8/eval"] = n(m(list(ahigh), 2.0))
    save("discriminators_t8", **out)


# ------------------------------------------------------------------- dataset-side sampler
def capture_sampling():
    """`sampling.farthest_point_sampling` (sampling.py:50-106, the numba FPS of the data loaders:
    train_utils.py:126, tempo_dataset.py:78, msr_dataset.py:94,130) run as plain numpy through the
    numba pass-through shim: indices for three clouds incl. exact duplicates and points at / near
    the origin (every point is eligible here, unlike pointnet2's FPS)."""
    import sampling as ref_sampling
    out = {}
    _, high = fluid_clip(1, 2048, 8, 1, seed=40)
    a = n(high[0][0]).astype(np.float32)
    _, ahigh = action_clip(1, 1024, 16, 1, seed=41)           # last eighth = exact repeats
    b = n(ahigh[0][0]).astype(np.float32)
    c = a[:512].copy()
    c[3] = 0.0                                                 # a point AT the origin
    c[100] = np.float32(1e-3) * c[100]                         # and one with |x|^2 << 1e-3
    c[200:204] = c[7]                                          # a run of duplicates
    for tag, pts, k, start in (("fluid", a, 256, 5), ("dup", b, 128, 0), ("origin", c, 64, 17)):
        idx, _ = ref_sampling.farthest_point_sampling(pts, k, initial_idx=start)
        out[f"{tag}/pts"], out[f"{tag}/k"], out[f"{tag}/start"] = pts, np.int64(k), np.int64(start)
        out[f"{tag}/idx"] = np.asarray(idx, dtype=np.int64)
        assert len(set(idx.tolist())) > k // 2
    save("sampling_fps", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gen", "dis", "loss", "steps", "sampling", "dis8"]
    if "sampling" in which:
        capture_sampling()
    if "dis8" in which:
        capture_discriminators_t8()
    if "gen" in which:
        capture_generator()
    if "dis" in which:
        capture_discriminators()
    if "loss" in which:
        capture_losses()
    if "steps" in which:
        for kind in ("fluid_keep", "fluid_init", "action"):
            step_fixture(kind)
