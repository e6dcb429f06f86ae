"""Segmented launches (nseg calls of one module in one pass) against the separate calls they stand for."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [0, 16, 256])          # 256: few long groups, walked as runs
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nseg,Ps", [(3, 2 * 64 * 16), (7, 512), (70, 256)])   # 7: lane groups of 8 with one idle; 70: two sweeps of the finalize kernels
def test_row_bn_act_segments_equal_separate_calls(K, dtype, nseg, Ps):
    import tpgan_amd.ops as ops
    torch.manual_seed(0)
    C = 64
    x = (torch.randn(nseg * Ps, C, device="cuda") * 1.5 + torch.arange(nseg, device="cuda").repeat_interleave(Ps)[:, None]).to(dtype)
    gamma = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device="cuda") * 0.1).requires_grad_(True)
    shift = torch.randn(C, device="cuda")
    rows = nseg * Ps // K if K else nseg * Ps
    gy = torch.randn(rows, C, device="cuda").to(dtype)

    def run(segmented):
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda")
        xx = x.clone().requires_grad_(True)
        g, b = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
        if segmented:
            y = ops.row_bn_act(xx, g, b, rm, rv, True, 0.1, 1e-5, 0.2, K, num_batches_tracked=nbt, nseg=nseg,
                               mean_shift=shift)
        else:
            y = torch.cat([ops.row_bn_act(xs, g, b, rm, rv, True, 0.1, 1e-5, 0.2, K, num_batches_tracked=nbt,
                                          mean_shift=shift) for xs in xx.chunk(nseg, 0)], 0)
        y.backward(gy)
        return y.detach(), xx.grad, g.grad, b.grad, rm, rv, int(nbt)

    a, b = run(True), run(False)
    assert a[6] == b[6] == nseg
    # same per-segment math; the segmented launch splits the rows over fewer workgroups per
    # segment, so the fp32 partial sums of the statistics associate differently (last-bit)
    tol = 1e-5 if dtype == torch.float32 else 1.6e-2                    # bf16: one rounding step
    assert torch.allclose(a[0].float(), b[0].float(), rtol=tol, atol=tol)
    assert torch.allclose(a[1].float(), b[1].float(), rtol=tol, atol=tol * float(b[1].float().abs().max()))
    assert torch.allclose(a[4], b[4], rtol=1e-5, atol=1e-6) and torch.allclose(a[5], b[5], rtol=1e-5, atol=1e-6)
    for u, v in zip(a[2:4], b[2:4]):                                    # dgamma / dbeta: fp64 sum vs fp32 adds
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-4 * float(v.abs().max()))
    # the shift reaches the running mean only
    rm0 = torch.zeros(C, device="cuda")
    for sg in range(nseg):
        rm0 = 0.9 * rm0 + 0.1 * (x[sg * Ps:(sg + 1) * Ps].float().mean(0) + shift)
    assert torch.allclose(a[4], rm0, atol=2e-3 if dtype == torch.bfloat16 else 1e-5)


def test_segmented_tail_equals_per_frame_calls():
    """T frames through a set-abstraction level as T segments == T separate forward_rows calls:
    features, BatchNorm running statistics, spectral-norm vectors and parameter gradients."""
    from tpgan_amd.set_abstraction import SSGSetConv
    torch.manual_seed(1)
    T, B, N = 3, 2, 512
    sa_a = SSGSetConv(npoint=128, radius=0.2, nsample=16, mlp=[3, 32, 32, 64], use_xyz=True, sn=True).cuda().train()
    sa_b = copy.deepcopy(sa_a)
    xyz = [torch.rand(B, N, 3, device="cuda") for _ in range(T)]
    feat = [x.clone().requires_grad_(True) for x in xyz]
    feat_b = [x.clone().requires_grad_(True) for x in xyz]
    pos_a, out_a = sa_a.forward_rows_frames(xyz, feat)                       # segmented tail
    outs_b = [sa_b.forward_rows(x, f) for x, f in zip(xyz, feat_b)]          # the reference's call pattern
    g = [torch.randn_like(o) for o in out_a]
    torch.autograd.backward(out_a, g)
    torch.autograd.backward([o[1] for o in outs_b], g)
    for t in range(T):
        assert torch.equal(pos_a[t], outs_b[t][0])
        assert torch.allclose(out_a[t], outs_b[t][1], rtol=1e-4, atol=1e-5)
        assert torch.allclose(feat[t].grad, feat_b[t].grad, rtol=1e-3, atol=1e-5 * float(feat_b[t].grad.abs().max() + 1))
    sd_a, sd_b = sa_a.state_dict(), sa_b.state_dict()
    for k in sd_a:
        assert torch.allclose(sd_a[k].float(), sd_b[k].float(), rtol=1e-5, atol=1e-6), k
    for (n, pa), (_, pb) in zip(sa_a.named_parameters(), sa_b.named_parameters()):
        if pb.grad is None:
            assert pa.grad is None, n
            continue
        if pa.grad is None:      # conv biases ahead of a BatchNorm: analytically zero gradient, folded away
            assert n.endswith("bias") and float(pb.grad.abs().max()) <= 1e-3 * float(g[0].abs().max()) * 64, n
            continue
        ref = float(pb.grad.abs().max()) + 1e-12
        assert float((pa.grad - pb.grad).abs().max()) <= 2e-3 * ref, n


def _no_dropout(m):
    for sub in m.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
    return m


def _compare_modules(ma, mb, outs_a, outs_b, rtol_out=2e-3, grad_rel=1e-2):
    for oa, ob in zip(outs_a, outs_b):
        oa, ob = oa.detach().float(), ob.detach().float()
        assert torch.allclose(oa, ob, rtol=rtol_out, atol=rtol_out * float(ob.abs().max() + 1e-6)), \
            float((oa - ob).abs().max())
    sd_a, sd_b = ma.state_dict(), mb.state_dict()
    for k in sd_a:                          # running statistics, spectral-norm vectors, counters
        # (the head's BatchNorm1d statistics sit behind every flow embedding and a batch of 4: the split spectral-norm
        # kernel evaluates W^T u as (W^T s) / |s| inside a chain of uses and from the stored u at the start of a call --
        # 1e-7 apart -- which reaches them as ~1e-5 at T = 8; logits are held at 2e-3 above)
        atol = 5e-5 if k.startswith("fc_layers") else 1e-5
        assert torch.allclose(sd_a[k].float(), sd_b[k].float(), rtol=1e-4, atol=atol), (k, float((sd_a[k].float() - sd_b[k].float()).abs().max()))
    ga = torch.cat([p.grad.reshape(-1) for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters())
                    if p.grad is not None and q.grad is not None])
    gb = torch.cat([q.grad.reshape(-1) for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters())
                    if p.grad is not None and q.grad is not None])
    assert float(gb.norm()) > 0
    rel = float((ga - gb).norm() / gb.norm())
    print("parameter gradients, relative L2 difference:", rel)
    assert rel <= grad_rel, rel
    for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        if p.grad is None and q.grad is not None:      # conv bias ahead of a BatchNorm: folded, ~zero gradient
            assert n.endswith("bias") and float(q.grad.abs().max()) <= 1e-3 * float(gb.abs().max()), n


def test_tempo_discriminator_forward_passes_equals_two_forwards():
    """fake + real batch as 2 x T segments of one pass == two successive forward calls (fp32)."""
    from tpgan_amd.set_abstraction import FluidTempoDis
    from tpgan_amd.synthetic import fluid_clip
    torch.manual_seed(2)
    dev = torch.device("cuda", 0)
    Da = _no_dropout(FluidTempoDis(3)).to(dev).train()
    Db = copy.deepcopy(Da)
    _, hi1 = fluid_clip(4, 1024, 8, 3, seed=1, device=dev)
    _, hi2 = fluid_clip(4, 1024, 8, 3, seed=2, device=dev)
    fa = [h.clone().requires_grad_(True) for h in hi1]
    fb = [h.clone().requires_grad_(True) for h in hi1]
    outs_a = Da.forward_passes([fa, hi2], 0.1)
    outs_b = [Db(fb, 0.1), Db(hi2, 0.1)]
    la = ((outs_a[0] - 0.1) ** 2).mean() + ((outs_a[1] - 1.0) ** 2).mean()
    lb = ((outs_b[0] - 0.1) ** 2).mean() + ((outs_b[1] - 1.0) ** 2).mean()
    la.backward(); lb.backward()
    _compare_modules(Da, Db, outs_a, outs_b)
    for x, y in zip(fa, fb):    # gradient reaching the generator's points (L2: a max-pool arg-max
        #                         flipping on a near-tie moves single entries by O(1))
        assert float((x.grad - y.grad).norm() / (y.grad.norm() + 1e-12)) <= 3e-2


@pytest.fixture(params=[False, True], ids=["sn_one_workgroup", "sn_split"])
def sn_split(request):
    """Both spectral-norm kernels.  The one-workgroup kernel evaluates a chain of uses exactly as successive calls do,
    so it isolates the SEGMENT logic under test; the split kernel (the default, csrc/spectral.hip) evaluates W^T u as
    (W^T s) / |s| inside a chain and from the stored u at the start of a call -- 1e-7 apart."""
    from tpgan_amd import ops
    be = ops.backend_for(torch.zeros(1, device="cuda"))
    prev = ops.SN_SPLIT[0]
    ops.SN_SPLIT[0] = request.param
    be._sn_plans.clear()
    yield request.param
    ops.SN_SPLIT[0] = prev
    be._sn_plans.clear()


@pytest.mark.parametrize("kind,T,N", [("fluid", 5, 2048), ("action", 8, 1024), ("fluid", 5, 16384)])
def test_long_clip_tempo_discriminators_forward_passes_equal_two_forwards(kind, T, N, sn_split):
    """The same equivalence at the clip lengths of cfg5 (FluidTempoDis(5): 10 flow embeddings, the depth-d conv used
    4 / 3 / 2 / 1 times per pass) and cfg4 (ActionTempoDis(8): 28) -- VERDICT r2 item 1(i): GPUTEST_r02's only
    diverging term was the temporal discriminator's update at T = 5.  Batch 4 (with 2 clips the head's BatchNorm1d is a
    sign network); outputs, BatchNorm running statistics, spectral-norm vectors, parameter and input gradients.  The
    last case is cfg5's own cloud size (16384 points: first-level inverted index beyond 64 KB of counters)."""
    from tpgan_amd.set_abstraction import ActionTempoDis, FluidTempoDis
    from tpgan_amd.synthetic import action_clip, fluid_clip
    torch.manual_seed(7)
    dev = torch.device("cuda", 0)
    B = 4
    if kind == "fluid":
        Da, R = _no_dropout(FluidTempoDis(T)).to(dev).train(), 0.10
        hi1, hi2 = (fluid_clip(B, N, 4, T, seed=s, device=dev)[1] for s in (1, 2))
    else:
        Da, R = _no_dropout(ActionTempoDis(T)).to(dev).train(), 2.0
        hi1, hi2 = (action_clip(B, N, 4, T, seed=s, device=dev)[1] for s in (1, 2))
    Db = copy.deepcopy(Da)
    fa = [h.clone().requires_grad_(True) for h in hi1]
    fb = [h.clone().requires_grad_(True) for h in hi1]
    outs_a = Da.forward_passes([fa, hi2], R)
    outs_b = [Db(fb, R), Db(hi2, R)]
    print(kind, T, "segments:", [o.flatten().tolist() for o in outs_a])
    print(kind, T, "separate:", [o.flatten().tolist() for o in outs_b])
    la = ((outs_a[0] - 0.1) ** 2).mean() + ((outs_a[1] - 1.0) ** 2).mean()
    lb = ((outs_b[0] - 0.1) ** 2).mean() + ((outs_b[1] - 1.0) ** 2).mean()
    la.backward(); lb.backward()
    # 10 / 28 flow embeddings with a max over 32 neighbours each: more arg-max near-ties that the different
    # association of the statistics' partial sums can flip than at T = 3.  The logits agree to 3e-4; the gradients'
    # relative L2 difference measured 1.6e-2 (library GEMMs) / 3.9e-2 (hand-written row-linear kernels) at T = 8 --
    # the same two evaluations, another rounding pattern: the figure is a property of the max-pool routing, the bound
    # (6e-2) is there for what it can catch: a segment run with the wrong weights or statistics is off by O(1).
    # With the split spectral-norm kernel the two evaluations also differ by 1e-7 in every weight: 7.2e-2 measured at
    # T = 8 -- the routing noise again, so that case gets 0.12 and the segment logic is held to 6e-2 by the other kernel.
    _compare_modules(Da, Db, outs_a, outs_b, grad_rel=0.12 if sn_split else 6e-2)
    for x, y in zip(fa, fb):
        assert float((x.grad - y.grad).norm() / (y.grad.norm() + 1e-12)) <= 6e-2


def test_spatial_discriminator_forward_passes_equals_two_forwards():
    from tpgan_amd.set_abstraction import FluidSpatialDis
    from tpgan_amd.synthetic import fluid_clip
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    Da = _no_dropout(FluidSpatialDis()).to(dev).train()
    Db = copy.deepcopy(Da)
    _, hi1 = fluid_clip(4, 2048, 8, 3, seed=1, device=dev)
    _, hi2 = fluid_clip(4, 2048, 8, 3, seed=2, device=dev)
    a, b = hi1[1].clone().requires_grad_(True), hi1[1].clone().requires_grad_(True)
    outs_a = Da.forward_passes([a, hi2[1]])
    outs_b = [Db(b), Db(hi2[1])]
    (((outs_a[0] - 0.1) ** 2).mean() + ((outs_a[1] - 1.0) ** 2).mean()).backward()
    (((outs_b[0] - 0.1) ** 2).mean() + ((outs_b[1] - 1.0) ** 2).mean()).backward()
    _compare_modules(Da, Db, outs_a, outs_b)
    assert float((a.grad - b.grad).norm() / (b.grad.norm() + 1e-12)) <= 3e-2


def _same_tensors(a, b):
    """Two nested plans hold bit-identical index tensors (and inverted indices where both have them)."""
    from tpgan_amd.set_abstraction import _plan_tensors
    if isinstance(a, dict):
        assert a.keys() == b.keys()
        return all(_same_tensors(a[key], b[key]) for key in a)
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        return all(_same_tensors(x, y) for x, y in zip(a, b))
    ta, tb = _plan_tensors(a), _plan_tensors(b)
    assert torch.equal(ta[0], tb[0])
    if len(ta) == len(tb) == 3:         # inverted index: deterministic since round 3 (entries ascending per bucket)
        assert torch.equal(ta[1], tb[1]) and torch.equal(ta[2], tb[2])
    return True


def test_joint_index_plans_equal_separate_index_plans():
    """index_plans (every search launched once for all passes) == one index_plan per pass, bit for bit,
    for both discriminators; and the merged plan is the merged plan."""
    from tpgan_amd.set_abstraction import FluidSpatialDis, FluidTempoDis, attach_plan_inverses
    from tpgan_amd.synthetic import fluid_clip
    dev = torch.device("cuda", 0)
    Dt, Ds = FluidTempoDis(3).to(dev), FluidSpatialDis().to(dev)
    clips = [fluid_clip(3, 2048, 8, 3, seed=s, device=dev)[1] for s in (1, 2, 3)]
    joint = Dt.index_plans(clips, 0.1)
    alone = [Dt.index_plan(c, 0.1) for c in clips]
    assert len(joint) == 3
    for j, a in zip(joint, alone):
        assert _same_tensors(j, a)
    assert _same_tensors(Dt.merge_plans(joint[:2]), Dt.merge_plans(alone[:2]))
    assert _same_tensors(attach_plan_inverses(joint[2]), alone[2])

    clouds = [c[1] for c in clips]
    joint, alone = Ds.index_plans(clouds), [Ds.index_plan(c) for c in clouds]
    for j, a in zip(joint, alone):
        assert _same_tensors(j, a)
    assert _same_tensors(Ds.merge_plans(joint[1:]), Ds.merge_plans(alone[1:]))
    assert _same_tensors(attach_plan_inverses(joint[0]), alone[0])
    # a plan of index_plans drives a forward like the plan of index_plan does
    Ds.eval()
    with torch.no_grad():
        assert torch.equal(Ds(clouds[0], plan=joint[0]), Ds(clouds[0], plan=alone[0]))


def test_prepared_spectral_norm_weights_equal_the_forwards_own():
    """sn_prepare + forward == forward (same power iterations, same outputs, same u / v buffers)."""
    from tpgan_amd.set_abstraction import FluidSpatialDis, sn_discard_prepared
    from tpgan_amd.synthetic import fluid_clip
    torch.manual_seed(5)
    dev = torch.device("cuda", 0)
    Da = _no_dropout(FluidSpatialDis()).to(dev).train()
    Db = copy.deepcopy(Da)
    _, hi = fluid_clip(4, 2048, 8, 3, seed=1, device=dev)
    Da.prepare_sn(1)
    Da.prepare_sn(2)                       # two forwards ahead: a single call, then a fake + real pass
    ya = [Da(hi[0])] + Da.forward_passes([hi[1], hi[2]])
    yb = [Db(hi[0])] + Db.forward_passes([hi[1], hi[2]])
    # (the power iteration's cross-workgroup sums are not bitwise reproducible run to run: two inline
    # forwards of two copies differ by the same 1e-7 in u / v and 1e-4 in the outputs)
    for a, b in zip(ya, yb):
        assert torch.allclose(a, b, rtol=0, atol=1e-3)
    for (n, a), (_, b) in zip(Da.named_buffers(), Db.named_buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-5), n
    sn_discard_prepared()
