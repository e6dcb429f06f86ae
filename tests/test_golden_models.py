"""Model-level parity against fixtures captured from the REFERENCE's own Python
(tests/golden/capture_goldens.py, run in the build container; the reference never travels).

Each test (1) builds this repo's module under the same torch seed as the reference module
was built and proves, via per-tensor checksums, that every state-dict entry has the same
name and bit-identical values; (2) runs it on the stored inputs; (3) compares with the
stored reference outputs.  On CPU the ops are served by the oracle (checker registered by
the `oracle_cpu` fixture); the `gpu`-marked twins run the same comparison through the HIP
kernels.  Tolerances: index-derived quantities exact; fp32 features 1e-5 on CPU (same conv
kernels as the reference run) and 2e-4 on the GPU (different GEMM summation order through
~20 layers; the op-level 1e-5 bar is enforced in test_ops_gpu.py).
"""
import contextlib
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# GPU bounds of the reference-order route (hipBLASLt / native conv summation order differs from
# the CPU run the goldens come from; training-mode BatchNorm on untrained nets multiplies absolute
# differences by up to 1/sqrt(eps) = 316): measured values are printed by the tests
GPU_TRAIN_TOL = 5e-3
# whole steps, on fixtures whose clip was selected for stability (capture_goldens.step_fixture): the six
# losses agree to ~2e-5 on every path (measured: fluid tempo_G 1.019007 reference / 1.019016 default order
# on CPU / 1.019018, 1.019023 on the GPU); held at 5e-4.  Parameters after the SGD step = gradients: a
# max-pool winner near a tie may differ between summation orders and moves single entries by lr * O(1),
# so the per-tensor checksums are held at 5e-3 wherever the arithmetic is not the reference's own
STEP_LOSS_TOL = 5e-4
STEP_STATE_TOL = 5e-3


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def check_weights(module, gold, prefix, atol=0.0, skip=None):
    sd = module.state_dict()
    keys = [k[len(prefix) + 1:] for k in gold.files if k.startswith(prefix + "/")]
    assert sorted(keys) == sorted(sd.keys()), set(keys) ^ set(sd.keys())
    for k in keys:
        if skip and k.startswith(skip):
            continue
        v = sd[k].detach().double().cpu()
        want = gold[f"{prefix}/{k}"]
        got = np.array([v.sum().item(), v.abs().sum().item(), float(v.numel())])
        assert np.allclose(got, want, rtol=0, atol=atol * max(1.0, want[1])), (k, got, want)


def close(a, b, tol):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert a.shape == b.shape and err <= tol * scale, (a.shape, b.shape, err, tol * scale)


def set_mask_head(net, mode, seed):
    last = net.filter_block.decoder[1]
    with torch.no_grad():
        if mode == "keep":
            last.weight.zero_()
            last.bias.fill_(1.0)
        elif mode == "mixed":
            g = torch.Generator().manual_seed(seed)
            last.weight.copy_(30.0 * torch.randn(last.weight.shape, generator=g))
            last.bias.fill_(-0.01)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@contextlib.contextmanager
def cpu_dropout():
    """Dropout masks drawn from torch's CPU generator, whatever device the activations are on.

    The goldens were captured from the reference on the CPU: its `nn.Dropout` draws
    `empty_like(x).bernoulli_(1 - p)` from the default CPU generator, interleaved with the step's
    `torch.randperm` draws (also CPU).  A GPU run draws dropout from the device's Philox stream
    instead, which made every quantity downstream of a head's Dropout incomparable.  With this
    patch the GPU run consumes the CPU generator exactly like the reference's run did, so ALL
    losses, BatchNorm statistics and post-step parameters can be compared on the GPU too."""
    orig = torch.nn.Dropout.forward

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        noise = torch.empty(x.shape, dtype=torch.float32).bernoulli_(1.0 - self.p).div_(1.0 - self.p)
        return x * noise.to(device=x.device, dtype=x.dtype)
    torch.nn.Dropout.forward = forward
    try:
        yield
    finally:
        torch.nn.Dropout.forward = orig


# ------------------------------------------------------------------------------ generator
def run_generator(dev, tol):
    from tpgan_amd.srnet import NoMaskSRNet, SRNet
    g = load("generator_srnet")
    torch.manual_seed(11)
    net = SRNet(3, 128)
    check_weights(net, g, "w")
    net = net.to(dev)
    x = _t(g["x"], dev)
    for mode in ("init", "keep", "mixed"):
        set_mask_head(net, mode, 3)
        for hard in (False, True):
            pos, mask, padded = net(x, x, hard_masking=hard)
            tag = f"{mode}/{'hard' if hard else 'soft'}"
            close(pos, g[f"{tag}/pos"], tol)
            close(mask, g[f"{tag}/mask"], tol)
            if f"{tag}/padded" in g.files:
                want = g[f"{tag}/padded"]
                if dev == "cpu":
                    close(padded, want, tol * 1e3 if mode == "mixed" else tol)
                else:
                    assert padded.shape[0] == want.shape[0]
            else:
                assert padded is None
    # the mixed regime really pads with 999 and the all-keep regime keeps everything
    assert (g["mixed/hard/padded"] == 999).any() and g["keep/hard/padded"].shape[1] == 8 * x.shape[1]
    torch.manual_seed(12)
    net6 = SRNet(6, 128)
    check_weights(net6, g, "w6")
    net6 = net6.to(dev)
    set_mask_head(net6, "keep", 0)
    pos, mask, _ = net6(_t(g["f6"], dev), x, hard_masking=False)
    close(pos, g["six/pos"], tol)
    close(mask, g["six/mask"], tol)
    set_mask_head(net, "keep", 0)
    hist = []
    p1, hist = net.forward_with_context(x[:1], x[:1], hist)
    p2, hist = net.forward_with_context(x[1:], x[1:], hist)
    close(p1, g["ctx/p1"], tol)
    close(p2, g["ctx/p2"], tol)
    # frames batched through the body == separate calls
    outs = net.forward_frames([x[:1], x[1:]], [x[:1], x[1:]], hard_masking=True)
    a, _, _ = net(x[:1], x[:1], hard_masking=True)
    close(outs[0][0], a.detach().cpu().numpy(), tol)

    g = load("generator_nomask")
    torch.manual_seed(13)
    net = NoMaskSRNet(3, 128, upsample_ratio=16)
    check_weights(net, g, "w")
    net = net.to(dev)
    pos, edge = net(_t(g["x"], dev), _t(g["x"], dev))
    close(pos, g["pos"], tol)
    close(edge, g["edge"], tol)


def test_generator_cpu(oracle_cpu):
    run_generator("cpu", 1e-5)


@pytest.mark.gpu
def test_generator_gpu():
    run_generator("cuda", 2e-4)


# -------------------------------------------------------------------------- discriminators
def run_discriminators(dev, tol, train_tol=None, state_tol=1e-5):
    from tpgan_amd import set_abstraction as SA
    train_tol = train_tol or tol
    g = load("discriminators")
    high = [_t(x, dev) for x in g["fluid"]]
    ahigh = [_t(x, dev) for x in g["action"]]
    specs = [("fluid_spatial", SA.FluidSpatialDis, lambda m: m(high[1])),
             ("fluid_tempo", lambda: SA.FluidTempoDis(3), lambda m: m(list(high), 0.10)),
             ("action_spatial", SA.ActionSpatialDis, lambda m: m(ahigh[1])),
             ("action_tempo", lambda: SA.ActionTempoDis(3), lambda m: m(list(ahigh), 2.0))]
    for i, (name, make, run) in enumerate(specs):
        torch.manual_seed(20 + i)
        m = make()
        check_weights(m, g, f"{name}/w")
        m = m.to(dev).train()
        torch.manual_seed(100 + i)              # same dropout draws as the reference run
        with cpu_dropout():                     # (CPU generator on either device)
            close(run(m), g[f"{name}/train"], train_tol)
        check_weights(m, g, f"{name}/w_after", atol=state_tol)       # BN stats + spectral-norm u/v
        m.eval()
        close(run(m), g[f"{name}/eval"], tol)
    # 999-padded clouds, seeded np.random replacement of dummy centres
    torch.manual_seed(24)
    m = SA.FluidSpatialDis().to(dev).eval()
    np.random.seed(77)
    close(m(_t(g["padded"], dev)), g["fluid_spatial/padded_eval"], tol)
    # five frames -> ten FlowEmbedding calls
    torch.manual_seed(25)
    m = SA.FluidTempoDis(5)
    check_weights(m, g, "fluid_tempo5/w")
    m = m.to(dev).eval()
    close(m([_t(x, dev) for x in g["fluid5"]], 0.10), g["fluid_tempo5/eval"], tol)
    # radius search + kNN fill (two searches in the reference) == one kNN search here
    idx = SA.ball_query_wrapper(0.04, 32, high[0][:, :256].contiguous(), high[1])
    assert np.array_equal(idx.cpu().numpy(), g["bqw/idx"])


def test_discriminators_reference_order_cpu(oracle_cpu):
    """Reference order of operations: fp32-rounding-level parity with the reference."""
    from tpgan_amd.set_abstraction import reference_order
    with reference_order():
        run_discriminators("cpu", 1e-5)


def test_discriminators_cpu(oracle_cpu):
    """Default MI355X order (first layer before the gather).  Algebraically identical; on these
    UNTRAINED nets the batch variance seen by training-mode BatchNorm is << eps = 1e-5, so BN
    multiplies absolute differences by up to 1/sqrt(eps) = 316: eval-mode logits and the
    BN / spectral-norm state stay at 1e-5-level, train-mode logits are held at 5e-3."""
    run_discriminators("cpu", 2e-5, train_tol=5e-3, state_tol=5e-5)


@pytest.mark.gpu
def test_discriminators_gpu():
    """Default order through the HIP kernels: eval logits and BN / spectral-norm state at 2e-4,
    train-mode logits at the CPU twin's 5e-3 (BatchNorm over variance << eps, see above)."""
    run_discriminators("cuda", 2e-4, train_tol=5e-3, state_tol=2e-4)


@pytest.mark.gpu
def test_discriminators_reference_order_gpu():
    """Reference order on the GPU: QueryAndGroup / grouping_operation / GroupAll
    inside the four discriminators, train AND eval logits, state after the train pass."""
    from tpgan_amd.set_abstraction import reference_order
    with reference_order():
        run_discriminators("cuda", 2e-4, train_tol=GPU_TRAIN_TOL, state_tol=2e-4)


def run_action_tempo8(dev, tol, train_tol, state_tol):
    """cfg4's temporal discriminator: eight frames, 28 FlowEmbedding calls at seven depths."""
    from tpgan_amd import set_abstraction as SA
    g = load("discriminators_t8")
    frames = [_t(x, dev) for x in g["action8"]]
    torch.manual_seed(60)
    m = SA.ActionTempoDis(8)
    check_weights(m, g, "action_tempo8/w")
    m = m.to(dev).train()
    torch.manual_seed(160)
    with cpu_dropout():
        close(m(list(frames), 2.0), g["action_tempo8/train"], train_tol)
    check_weights(m, g, "action_tempo8/w_after", atol=state_tol)
    m.eval()
    close(m(list(frames), 2.0), g["action_tempo8/eval"], tol)


def test_action_tempo8_reference_order_cpu(oracle_cpu):
    from tpgan_amd.set_abstraction import reference_order
    with reference_order():
        run_action_tempo8("cpu", 1e-5, 1e-5, 1e-5)


def test_action_tempo8_cpu(oracle_cpu):
    """Default order.  Train-mode logits at 2e-2: the head's BatchNorm1d normalises over the TWO
    clips of the batch (measured batch variance down to 1.5e-6 < eps, i.e. up to a 1/sqrt(eps) = 316x
    gain on the 1e-6-level differences of the two algebraically equal orders after seven chained
    flow-embedding depths); the running statistics of the head's second BatchNorm sit downstream of
    that and are held at 5e-4; eval logits (they use those running statistics) at 2e-4."""
    run_action_tempo8("cpu", 2e-4, 2e-2, 5e-4)


@pytest.mark.gpu
def test_action_tempo8_gpu():
    run_action_tempo8("cuda", 2e-4, 2e-2, 5e-4)


@pytest.mark.gpu
def test_action_tempo8_reference_order_gpu():
    from tpgan_amd.set_abstraction import reference_order
    with reference_order():
        run_action_tempo8("cuda", 2e-4, 2e-2, 5e-4)


# --------------------------------------------------------------------------------- losses
def run_losses(dev, tol):
    from tpgan_amd import losses
    g = load("losses")
    low, high = _t(g["low"], dev), _t(g["high"], dev)
    for n_iter in (5, 12):
        pred = _t(g["pred"], dev).requires_grad_(True)
        mask = _t(g["mask"], dev).requires_grad_(True)
        total, cd, ml = losses.tpugan_sr_loss(100., high, pred, low, mask, 0.025, n_iter)
        close(total, g[f"it{n_iter}/total"], tol)
        close(cd, g[f"it{n_iter}/cd"], tol)
        close(ml, g[f"it{n_iter}/ml"], tol)
        gp, gm = torch.autograd.grad(total, [pred, mask], allow_unused=True)
        close(gp, g[f"it{n_iter}/grad_pred"], tol)
        if f"it{n_iter}/grad_mask" in g.files:
            close(gm, g[f"it{n_iter}/grad_mask"], tol)
        else:
            assert gm is None
    close(losses.chamfer_distance_loss(high[0], _t(g["pred"], dev)[0]), g["cd_unbatched"], tol)


def test_losses_cpu(oracle_cpu):
    run_losses("cpu", 1e-5)


@pytest.mark.gpu
def test_losses_gpu():
    run_losses("cuda", 1e-5)


# ----------------------------------------------------------------------------- train steps
def run_step(kind, dev, tol, gan_tol=None, state_tol=2e-5):
    from tpgan_amd import set_abstraction as SA
    gan_tol = gan_tol or tol
    from tpgan_amd.gan_step import tempo_gan_step, tempo_gan_step_no_mask
    from tpgan_amd.srnet import NoMaskSRNet, SRNet
    g = load(f"step_{kind}")
    if kind == "action":
        torch.manual_seed(40)
        G = NoMaskSRNet(3, 128, upsample_ratio=16)
        torch.manual_seed(41)
        Ds = SA.ActionSpatialDis()
        torch.manual_seed(42)
        Dt = SA.ActionTempoDis(3)
        opt = Namespace(R=2.0, w=2.0)
    else:
        torch.manual_seed(30)
        G = SRNet(3, 128)
        torch.manual_seed(31)
        Ds = SA.FluidSpatialDis()
        torch.manual_seed(32)
        Dt = SA.FluidTempoDis(3)
        if kind == "fluid_keep":
            set_mask_head(G, "keep", 0)
        opt = Namespace(use_vel=False, in_node_feats=3, cutoff=0.025, R=0.10, w=0.5)
    for tag, m in (("G", G), ("Ds", Ds), ("Dt", Dt)):
        check_weights(m, g, f"w/{tag}")
        m.to(dev)
    low = [_t(x, dev) for x in g["low"]]
    high = [_t(x, dev) for x in g["high"]]
    og = torch.optim.SGD(G.parameters(), lr=0.05)
    ot = torch.optim.SGD(Dt.parameters(), lr=0.05)
    os_ = torch.optim.SGD(Ds.parameters(), lr=0.05)
    torch.manual_seed(500)
    np.random.seed(500)
    # the reference's draws on either device: np.random and torch.randperm are host generators
    # already; the heads' dropout masks come from the CPU generator through cpu_dropout()
    with cpu_dropout():
        if kind == "action":
            losses = tempo_gan_step_no_mask(G, Ds, Dt, low, high, opt, 12, og, ot, os_)
        else:
            losses = tempo_gan_step(G, Ds, Dt, low, None, high, None, 1.0, opt, 12, og, ot, os_)
    want = {k[5:]: float(g[k]) for k in g.files if k.startswith("loss/")}
    assert set(losses) == set(want)
    print(kind, dev, {k: (round(losses[k], 6), round(want[k], 6)) for k in want})
    for k in want:
        t = tol if k in ("Chamfer_distance_no_norm", "masking_loss") else gan_tol
        assert abs(losses[k] - want[k]) <= t * max(1.0, abs(want[k])), (k, losses[k], want[k])
    # parameters after one SGD step == reference's, i.e. the gradients agree
    for tag, m in (("G", G), ("Ds", Ds), ("Dt", Dt)):
        check_weights(m, g, f"w_after/{tag}", atol=state_tol)
    if kind == "fluid_init":
        assert losses["tempo_G_loss"] == 0.0 and losses["spatial_D_loss"] == 0.0   # gate closed


@pytest.mark.parametrize("kind", ["fluid_keep", "fluid_init", "action"])
def test_train_step_reference_order_cpu(oracle_cpu, kind):
    """Losses and post-step parameters (= gradients) at fp32-rounding level of the reference."""
    from tpgan_amd.set_abstraction import reference_order
    with reference_order():
        run_step(kind, "cpu", 2e-5)


@pytest.mark.parametrize("kind", ["fluid_keep", "fluid_init", "action"])
def test_train_step_cpu(oracle_cpu, kind):
    """Default MI355X order (first layer before the gather): ALL six losses and the parameters after
    the step against the reference's.  This works because the fixtures' clips were selected for
    stability (capture_goldens.step_fixture): on an arbitrary clip an untrained step sits on discrete
    decisions (FPS picks, ball-query membership, max-pool winners on the generated clouds) that
    rounding-level differences flip -- the first fixture of this repo read tempo_G_loss = 0.96814 with
    8 oneDNN threads and 0.94807 with one thread, with oneDNN off, and on every path of this repo."""
    run_step(kind, "cpu", 2e-5, gan_tol=STEP_LOSS_TOL, state_tol=STEP_STATE_TOL)


def test_both_orders_give_the_same_gradients(oracle_cpu):
    """Rows order (first layer before the gather) vs the reference's order, same module, same
    inputs, BatchNorm in eval mode (well conditioned): outputs and ALL gradients agree."""
    from tpgan_amd import set_abstraction as SA
    from tpgan_amd.graph_conv import reference_order
    from tpgan_amd.srnet import SRNet
    g = load("discriminators")
    high = [_t(x, "cpu") for x in g["fluid"]]

    def grads(make, run, seed):
        out = []
        for ref in (True, False):
            torch.manual_seed(seed)
            m = make()
            if hasattr(m, "fc_layers"):
                # compare the pooled features: the head's BatchNorm1d normalises over the TWO
                # clips of the batch (variance << eps) and would only add a 316x error gain
                m.fc_layers = torch.nn.Identity()
            # calibrate BN running statistics with one training-mode pass (momentum 1 => running
            # stats := batch stats), identically for both instances, then evaluate: activations
            # stay O(1) and BN is well conditioned
            for mod in m.modules():
                if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    mod.momentum = 1.0
            with reference_order(True), torch.no_grad():
                run(m.train(), [h.clone() for h in high])
            m.eval()
            xs = [h.clone().requires_grad_(True) for h in high]
            with reference_order(ref):
                y = run(m, xs)
            y.sum().backward()
            out.append((y.detach(), [p.grad for p in m.parameters()], [x.grad for x in xs if x.grad is not None]))
        return out

    cases = [(SA.FluidSpatialDis, lambda m, xs: m(xs[1]), 1),
             (lambda: SA.FluidTempoDis(3), lambda m, xs: m(xs, 0.10), 2),
             (lambda: SRNet(3, 128), lambda m, xs: m(xs[0][:, ::8], xs[0][:, ::8])[0], 3)]
    for make, run, seed in cases:
        (y0, p0, x0), (y1, p1, x1) = grads(make, run, seed)
        close(y1, y0.numpy(), 5e-4)
        assert len(p0) == len(p1) and len(x0) == len(x1) and len(x0) >= 1
        for a, b in zip(p0 + x0, p1 + x1):
            assert (a is None) == (b is None)
            if a is not None:
                # max-pool routes gradients through arg-max winners and a near-tie may flip
                # between the two orders, which moves a few entries by O(1): compare in L2
                rel = float((a - b).norm() / a.norm().clamp_min(1e-6))
                assert rel <= 2e-2, rel


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fluid_keep", "fluid_init", "action"])
def test_train_step_gpu(kind):
    """Default MI355X order on the GPU -- the path bench.py times, in fp32 -- with the reference's
    draws (cpu_dropout): all six losses and the parameters after one SGD step against the goldens."""
    run_step(kind, "cuda", 2e-4, gan_tol=STEP_LOSS_TOL, state_tol=STEP_STATE_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fluid_keep", "fluid_init", "action"])
def test_train_step_reference_order_gpu(kind):
    """The zero-edit integration route on the GPU: reference order of operations (ball query ->
    grouping_operation -> conv on grouped positions: `QueryAndGroup`, `_Group` autograd Functions
    over tpg_group_fwd/bwd, what discriminator.py:190,270-273 would call), the reference's host
    draws, ALL six losses and the parameters after one SGD step against the goldens."""
    from tpgan_amd.set_abstraction import reference_order
    with reference_order():
        run_step(kind, "cuda", 2e-4, gan_tol=STEP_LOSS_TOL, state_tol=STEP_STATE_TOL)


def test_use_vel_step_runs_on_the_oracle_backend(oracle_cpu):
    """`--use_vel` (train_step_final.py:96-104,147-150,171-181): advection features through the fused
    interpolation, generator input [pos | vel*DT] with in_node_feats = 6.  No golden exists (the
    reference needs DGL for this path: PARITY UNPINNED); the interpolation itself is checked against
    the reference's algorithm in tests/test_oracle_cpu.py."""
    import numpy as np
    import torch
    from argparse import Namespace
    from tpgan_amd.gan_step import tempo_gan_step
    from tpgan_amd.set_abstraction import FluidSpatialDis, FluidTempoDis
    from tpgan_amd.srnet import SRNet
    from tpgan_amd.synthetic import fluid_clip, force_all_keep
    torch.manual_seed(0); np.random.seed(0)
    G = force_all_keep(SRNet(6, 128))
    Ds, Dt = FluidSpatialDis(), FluidTempoDis(3)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-4) for m in (G, Dt, Ds)]
    low, high, lvel, hvel = fluid_clip(2, 1024, 8, 3, seed=3, with_vel=True)
    opt = Namespace(use_vel=True, in_node_feats=6, cutoff=0.025, R=0.10, w=0.5)
    before = [p.detach().clone() for p in Dt.parameters()]
    out = tempo_gan_step(G, Ds, Dt, low, lvel, high, hvel, 1.0, opt, 12, opts[0], opts[1], opts[2], force_gate=True)
    assert all(np.isfinite(v) for v in out.values()) and out["tempo_D_loss"] > 0
    assert any(not torch.equal(a, b) for a, b in zip(before, Dt.parameters()))
    # the discriminator really saw the advection features: same seeds without them -> another loss
    torch.manual_seed(0); np.random.seed(0)
    G2 = force_all_keep(SRNet(6, 128))
    Ds2, Dt2 = FluidSpatialDis(), FluidTempoDis(3)
    opts2 = [torch.optim.Adam(m.parameters(), lr=1e-4) for m in (G2, Dt2, Ds2)]
    hvel0 = [torch.zeros_like(v) for v in hvel]
    out0 = tempo_gan_step(G2, Ds2, Dt2, low, lvel, high, hvel0, 1.0, opt, 12, opts2[0], opts2[1], opts2[2],
                          force_gate=True)
    assert abs(out0["tempo_G_loss"] - out["tempo_G_loss"]) > 1e-6


@pytest.mark.gpu
def test_use_vel_step_runs_on_the_gpu():
    import numpy as np
    import torch
    from argparse import Namespace
    from tpgan_amd.gan_step import tempo_gan_step
    from tpgan_amd.set_abstraction import FluidSpatialDis, FluidTempoDis
    from tpgan_amd.srnet import SRNet
    from tpgan_amd.synthetic import fluid_clip, force_all_keep
    dev = torch.device("cuda", 0)
    torch.manual_seed(0); np.random.seed(0)
    G = force_all_keep(SRNet(6, 128)).to(dev)
    Ds, Dt = FluidSpatialDis().to(dev), FluidTempoDis(3).to(dev)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-4) for m in (G, Dt, Ds)]
    low, high, lvel, hvel = fluid_clip(2, 1024, 8, 3, seed=3, with_vel=True, device=dev)
    opt = Namespace(use_vel=True, in_node_feats=6, cutoff=0.025, R=0.10, w=0.5)
    for n_iter in (12, 13):
        out = tempo_gan_step(G, Ds, Dt, low, lvel, high, hvel, 1.0, opt, n_iter, opts[0], opts[1], opts[2],
                             force_gate=True, amp_dtype=torch.bfloat16)
        assert all(np.isfinite(v) for v in out.values())
        assert (out["tempo_D_loss"] > 0) == (n_iter % 2 == 0)
