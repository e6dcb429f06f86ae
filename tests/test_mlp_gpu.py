"""The fused MFMA tail-layer kernels (csrc/mlp_fused.hip) against a plain PyTorch fp32 statement of
the same op on the same (bf16-rounded) inputs.

Tolerances: the kernel multiplies bf16 operands exactly and accumulates in fp32 like the reference
product; the stored y is rounded once to bf16, so |y - y_ref| <= one bf16 ulp of |y_ref| (2^-8
relative) plus the fp32 summation-order noise; statistics are computed from the kernel's OWN
stored values and must match an fp64 recomputation from them to 1e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import tpgan_amd.ops as ops
    return ops.backend_for(torch.zeros(1, device="cuda"))


def _layer_inputs(P, Cin, Cout, nseg, seed, with_bn=True):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(P, Cin, generator=g) * 0.7 + 0.3 * torch.randn(1, Cin, generator=g)).bfloat16().cuda()
    W = (torch.randn(nseg, Cout, Cin, generator=g) / Cin ** 0.5).cuda()
    if with_bn:
        ss = torch.stack([torch.rand(nseg, Cin, generator=g) + 0.5, 0.3 * torch.randn(nseg, Cin, generator=g)], 1).cuda()
    else:
        ss = None
    return x, W, ss


def _ref_fwd(x, W, ss, slope, nseg):
    P = x.shape[0] // nseg
    ys = []
    for s in range(nseg):
        a = x[s * P:(s + 1) * P].float()
        if ss is not None:
            # one rounding of the exact x*sc + sh, like the kernel's fma
            a = (a.double() * ss[s, 0].double() + ss[s, 1].double()).float()
        a = torch.where(a > 0, a, a * slope).bfloat16().float()
        ys.append(a @ W[s].bfloat16().float().t())
    return torch.cat(ys, 0)


@pytest.mark.parametrize("Cin,Cout", [(64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (256, 128), (256, 256)])
@pytest.mark.parametrize("P,nseg", [(8 * 128 * 32, 1), (3 * 4096, 3), (1000, 1), (2 * 16 * 7, 2)])
def test_mlp_fwd_matches_torch(hip, Cin, Cout, P, nseg):
    x, W, ss = _layer_inputs(P, Cin, Cout, nseg, seed=Cin + Cout + P)
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    gamma, beta = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    y, ci = hip.mlp_fwd(x, ss, 0.01, W, nseg, 1e-5, 0.1, rm, rv, nbt, None, gamma, beta)
    mean, rstd = ci[:, 2], ci[:, 3]
    ref = _ref_fwd(x, W, ss, 0.01, nseg)
    err = (y.float() - ref).abs()
    # same bf16 operands, fp32 accumulation in another order, ONE rounding of y to bf16: within one
    # bf16 ulp (2^-7 relative: a sum that lands next to a rounding tie may go to the other neighbour)
    bound = ref.abs() * 2.0 ** -7 + 1e-5 * ref.abs().max()
    assert bool((err <= bound).all()), float((err - bound).max())
    # statistics of the stored tensor, per segment
    Ps = P // nseg
    yd = y.double().view(nseg, Ps, Cout)
    m_ref, v_ref = yd.mean(1), yd.var(1, unbiased=False)
    assert torch.allclose(mean.double(), m_ref, rtol=0, atol=1e-5 * float(m_ref.abs().max() + v_ref.sqrt().max()))
    assert torch.allclose(rstd.double(), (v_ref + 1e-5).rsqrt(), rtol=1e-5, atol=0)
    a = gamma * rstd
    assert torch.allclose(ci[:, 0], a) and torch.allclose(ci[:, 1], beta - mean * a, atol=1e-6)
    # running statistics chained segment after segment, unbiased variance, counter += nseg
    erm, erv = torch.zeros(Cout, dtype=torch.float64), torch.ones(Cout, dtype=torch.float64)
    for s in range(nseg):
        erm = 0.9 * erm + 0.1 * m_ref[s].cpu()
        erv = 0.9 * erv + 0.1 * (v_ref[s].cpu() * Ps / (Ps - 1))
    assert torch.allclose(rm.double().cpu(), erm, atol=1e-5) and torch.allclose(rv.double().cpu(), erv, rtol=1e-5, atol=1e-6)
    assert int(nbt) == nseg


def test_mlp_fwd_statistics_survive_a_large_mean(hip):
    """|mean| / sigma = 1e3 in the OUTPUT: the pivoted, Chan-combined sums keep the variance."""
    P, Cin, Cout = 65536, 64, 128
    x = torch.ones(P, Cin).bfloat16().cuda()
    x[:, 0] = (torch.randn(P) * 0.01).bfloat16().cuda()             # the only varying input
    W = torch.zeros(1, Cout, Cin).cuda()
    W[0, :, 1] = 8.0                                                  # constant part: 8
    W[0, :, 0] = 1.0                                                  # varying part: sigma 0.01
    y, ci = hip.mlp_fwd(x, None, 1.0, W, 1, 1e-5, 0.0, None, None, None, None, None, None)
    mean, rstd = ci[:, 2], ci[:, 3]
    yd = y.double()
    assert torch.allclose(mean.double()[0], yd.mean(0), atol=1e-6)
    assert torch.allclose(rstd.double()[0], (yd.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)


def test_mlp_fwd_is_bitwise_reproducible(hip):
    x, W, ss = _layer_inputs(6 * 8192, 64, 128, 3, seed=5)
    a = hip.mlp_fwd(x, ss, 0.2, W, 3, 1e-5, 0.1, None, None, None, None, None, None)
    b = hip.mlp_fwd(x, ss, 0.2, W, 3, 1e-5, 0.1, None, None, None, None, None, None)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # without statistics (a tail without BatchNorm): the same rows, no finalize
    y2, none = hip.mlp_fwd(x, ss, 0.2, W, 3, stats=False)
    assert none is None and torch.equal(y2, a[0])


# ------------------------------------------------------------------ the whole tail, forward + backward
def _ste_bf16(t):
    """bf16 rounding with a straight-through gradient: the reference rounds where the kernels round."""
    return t + (t.bfloat16().float() - t).detach()


def _ref_tail(x0, gammas, betas, Ws, slopes, K, nseg, eps=1e-5):
    outs = []
    P = x0.shape[0] // nseg
    for s in range(nseg):
        a = x0[s * P:(s + 1) * P]
        for l in range(len(gammas)):
            m, v = a.mean(0), a.var(0, unbiased=False)
            z = (a - m) * (gammas[l] * (v + eps).rsqrt()) + betas[l]
            a = torch.where(z > 0, z, z * slopes[l])
            if l < len(Ws):
                w = Ws[l][s] if Ws[l].dim() == 3 else Ws[l]
                a = _ste_bf16(_ste_bf16(a) @ _ste_bf16(w).t())       # operands and the stored product in bf16
        outs.append(a.view(P // K, K, -1).max(1)[0])
    return torch.cat(outs, 0)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))


@pytest.mark.parametrize("chain,K,P,nseg,per_seg_w", [
    ((64, 128), 32, 8 * 64 * 32, 1, False), ((64, 128), 32, 2 * 8 * 64 * 32, 2, True),
    ((128, 128), 32, 4096, 1, False), ((128, 256), 16, 3 * 2048, 3, True),
    ((256, 128, 256), 32, 2 * 4096, 2, True), ((256, 256, 256), 32, 4096, 1, False),
    ((64, 64, 128), 64, 4096, 1, False), ((128, 64), 8, 1000 * 8, 1, False),
    ((64, 128), 16, 7 * 320, 7, True), ((64, 128, 128), 16, 70 * 256, 70, True)])     # 70 segments: two sweeps of the finalize kernels
def test_mlp_tail_forward_backward(chain, K, P, nseg, per_seg_w):
    import tpgan_amd.ops as ops
    torch.manual_seed(sum(chain) + K)
    L = len(chain) - 1
    x0 = (torch.randn(P, chain[0]) * 0.8 + 0.2 * torch.randn(1, chain[0])).bfloat16().cuda()
    bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in chain]
    for bn in bns:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
    Ws = [(torch.randn(*((nseg,) if per_seg_w else ()), chain[l + 1], chain[l]) / chain[l] ** 0.5).cuda().requires_grad_(True)
          for l in range(L)]
    slopes = [0.01] * (L + 1)
    assert ops.mlp_tail_supported(x0, chain, K)
    xk = x0.clone().requires_grad_(True)
    out = ops.mlp_tail(xk, bns, Ws, slopes, K, nseg)
    gout = torch.randn(out.shape, device="cuda").bfloat16()
    params = [xk] + Ws + [p for bn in bns for p in (bn.weight, bn.bias)]
    got = torch.autograd.grad(out, params, gout)
    xr = x0.float().clone().requires_grad_(True)
    ref = _ref_tail(xr, [bn.weight for bn in bns], [bn.bias for bn in bns], Ws, slopes, K, nseg)
    want = torch.autograd.grad(ref, [xr] + Ws + [p for bn in bns for p in (bn.weight, bn.bias)], gout.float())
    # forward: the stored output is one bf16 rounding of a value that went through L bf16 products
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-2, err
    # (a max-pool winner may differ where two rows are within rounding: compared in L2)
    names = ["dx0"] + [f"dW{l + 1}" for l in range(L)] + [f"{n}{l}" for l in range(L + 1) for n in ("dgamma", "dbeta")]
    for name, g, w in zip(names, got, want):
        assert g.shape == w.shape, (name, g.shape, w.shape)
        rel = _rel(g.float(), w)
        assert rel <= 3e-2, (name, rel)
    # running statistics moved like nn.BatchNorm's own forward (momentum 0.1, nseg updates)
    assert all(int(bn.num_batches_tracked) == nseg for bn in bns)
    assert all(not torch.equal(bn.running_mean, torch.zeros_like(bn.running_mean)) for bn in bns)


def test_mlp_tail_frozen_weights_need_no_weight_gradient():
    """The generator step back-propagates through FROZEN discriminators: data gradient only."""
    import tpgan_amd.ops as ops
    torch.manual_seed(0)
    x0 = torch.randn(4096, 64).bfloat16().cuda().requires_grad_(True)
    bns = [torch.nn.BatchNorm1d(c).cuda().train().requires_grad_(False) for c in (64, 128)]
    W = (torch.randn(128, 64) / 8).cuda()
    out = ops.mlp_tail(x0, bns, [W], [0.01, 0.01], 32)
    (gx,) = torch.autograd.grad(out, [x0], torch.randn_like(out))
    assert torch.isfinite(gx).all() and float(gx.abs().max()) > 0


def test_mlp_tail_is_bitwise_reproducible():
    import tpgan_amd.ops as ops
    torch.manual_seed(1)
    x0 = torch.randn(3 * 8192, 64).bfloat16().cuda()
    W = (torch.randn(3, 128, 64) / 8).cuda().requires_grad_(True)
    res = []
    for _ in range(2):
        bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in (64, 128)]
        xk = x0.clone().requires_grad_(True)
        out = ops.mlp_tail(xk, bns, [W], [0.01, 0.01], 32, 3)
        g = torch.autograd.grad(out, [xk, W, bns[0].weight, bns[1].bias], torch.ones_like(out))
        res.append((out,) + tuple(g) + (bns[1].running_var.clone(),))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("C,K,S,nseg", [(64, 32, 256, 1), (64, 12, 300, 2), (128, 7, 96, 3)])
def test_tail_hands_its_row_sums_to_the_gather_that_fed_it(C, K, S, nseg):
    """A set-abstraction level is row_combine(ROW_SUB) -> fused tail.  The tail's backward sums its input gradient
    over each group of K rows while writing it (tpg_mlp_bn_bwd_apply_rowsum) and the gather's backward takes those
    sums as the gradient of Q instead of reading the rows again: same bits as the two-pass route, and a gradient
    that is NOT the tail's own tensor (here: accumulated with a second consumer's) must not pick the sums up."""
    import tpgan_amd.ops as ops
    torch.manual_seed(C + K)
    B, N = 2 * nseg, 500
    U0 = torch.randn(B, N, C, device="cuda")
    Q0 = torch.randn(B, S, C, device="cuda")
    idx = torch.randint(0, N, (B, S, K), device="cuda", dtype=torch.int32)
    W = (torch.randn(128, C) / C ** 0.5).cuda().requires_grad_(True)
    res = {}          # (the inverted index keeps entry order: the scatter sums are reproducible, bits can be compared)
    for tag, handoff, second in (("two-pass", False, False), ("handed", True, False), ("shared", True, True)):
        prev = ops.set_rowsum_handoff(handoff)
        try:
            bns = [torch.nn.BatchNorm1d(c).cuda().train() for c in (C, 128)]
            U, Q = U0.clone().requires_grad_(True), Q0.clone().requires_grad_(True)
            y = ops.row_combine(U, Q, idx, ops.ROW_SUB, out_dtype=torch.bfloat16)
            out = ops.mlp_tail(y.view(-1, C), bns, [W], [0.01, 0.01], K, nseg)
            loss = (out.float() * torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)).sum()
            if second:
                loss = loss + y.float().square().sum() * 1e-3
            res[tag] = torch.autograd.grad(loss, [U, Q])
            assert ops._ROWSUM_SLOT[0] is None or second        # taken (and emptied) by the gather's backward
        finally:
            ops.set_rowsum_handoff(prev)
    for a, b in zip(res["two-pass"], res["handed"]):
        assert torch.equal(a, b)
    # the shared case went the ordinary way: finite, and different from the tail-only gradient
    assert all(torch.isfinite(t).all() for t in res["shared"])
    assert not torch.equal(res["shared"][1], res["handed"][1])


# ------------------------------------------------------------------ EdgeConv MLP: the tail without BatchNorm
@pytest.mark.parametrize("chain,K,P", [((64, 64, 128), 20, 512 * 20 * 6), ((128, 128, 256), 12, 512 * 12 * 4),
                                         ((128, 128, 256), 4, 1000 * 4), ((64, 128), 9, 9 * 333)])
def test_mlp_tail_plain_forward_backward(chain, K, P):
    """ops.mlp_tail_plain (identity statistics) against the PyTorch statement of the EdgeConv MLP
    (gcn_lib/pointnet/gcn.py:207-211): [conv -> LeakyReLU(0.2)]* -> max over the k neighbours."""
    import tpgan_amd.ops as ops
    torch.manual_seed(sum(chain) + K)
    L = len(chain) - 1
    x0 = torch.randn(P, chain[0]).bfloat16().cuda()
    Ws = [(torch.randn(chain[l + 1], chain[l]) / chain[l] ** 0.5).cuda().requires_grad_(True) for l in range(L)]
    slopes = [1.0] + [0.2] * L
    xk = x0.clone().requires_grad_(True)
    out = ops.mlp_tail_plain(xk, Ws, slopes, K)
    gout = torch.randn(out.shape, device="cuda").bfloat16()
    got = torch.autograd.grad(out, [xk] + Ws, gout)
    xr = x0.float().clone().requires_grad_(True)
    a = xr
    for l in range(L):
        a = _ste_bf16(_ste_bf16(a) @ _ste_bf16(Ws[l]).t())
        a = torch.where(a > 0, a, a * 0.2)
    ref = a.view(P // K, K, -1).max(1)[0]
    want = torch.autograd.grad(ref, [xr] + Ws, gout.float())
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-2, err
    for name, g, w in zip(["dx0"] + [f"dW{l + 1}" for l in range(L)], got, want):
        assert g.shape == w.shape, (name, g.shape, w.shape)
        rel = _rel(g.float(), w)
        assert rel <= 3e-2, (name, rel)


def test_edgeconv_fused_tail_equals_the_separate_launches():
    """EdgeConv.forward_rows with the fused MFMA tail against the hipBLASLt + activation launches it
    replaces, bf16 autocast, same module: outputs and all gradients."""
    from tpgan_amd import graph_conv
    from tpgan_amd.graph_conv import EdgeConv
    torch.manual_seed(4)
    m = EdgeConv(64, 256, k=12, aggregate="max", mlp_layer=True, bn=False, insn=False).cuda()
    x = torch.randn(6, 512, 64, device="cuda")
    res = []
    for fused in (True, False):
        graph_conv.FUSED_EDGE_TAILS[0] = fused
        try:
            xi = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m.forward_rows(xi)
            g = torch.autograd.grad(y.float().square().sum(), [xi] + list(m.parameters()))
            res.append((y.float(),) + tuple(t.float() for t in g))
        finally:
            graph_conv.FUSED_EDGE_TAILS[0] = True
    for a, b in zip(*res):
        assert a.shape == b.shape and _rel(a, b) <= 3e-2, _rel(a, b)


# ------------------------------------------------------------ the 16 -> 16 -> 32 tail of the IDGCN EdgeConvs
def _small_tail_reference(h, W1, W2, s1, s2, K):
    z2 = F.leaky_relu(h.float() @ W1.t(), s1) @ W2.t()
    P = h.shape[0] // K
    return F.leaky_relu(z2.view(P, K, -1).max(1)[0], s2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("P,K", [(12288, 20), (12288, 10), (1000, 9), (37, 3), (5, 1)])
def test_small_tail_forward_backward(dtype, P, K):
    """csrc/mlp_small.hip against plain PyTorch fp32 on the same rows: output, gradient to the rows, both
    weight gradients.  The kernels compute in fp32 whatever the row type: fp32 rows to 1e-5 (north_star's
    tolerance), bf16 rows to the rounding of the stored output / gradient (2^-8)."""
    from tpgan_amd import ops
    torch.manual_seed(P + K)
    h = torch.randn(P * K, 16, device="cuda").to(dtype)
    W1 = (torch.randn(16, 16, device="cuda") * 0.3).requires_grad_(True)
    W2 = (torch.randn(32, 16, device="cuda") * 0.3).requires_grad_(True)
    g = torch.randn(P, 32, device="cuda").to(dtype)
    hi = h.clone().requires_grad_(True)
    y = ops.small_tail(hi, W1, W2, 0.2, 0.2, K)
    assert y.dtype == dtype and y.shape == (P, 32)
    gh, g1, g2 = torch.autograd.grad(y, [hi, W1, W2], g)
    hr = h.float().clone().requires_grad_(True)
    yr = _small_tail_reference(hr, W1, W2, 0.2, 0.2, K)
    rh, r1, r2 = torch.autograd.grad(yr, [hr, W1, W2], g.float())
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    assert _rel(y.float(), yr) <= tol, _rel(y.float(), yr)
    assert _rel(gh.float(), rh) <= tol, _rel(gh.float(), rh)
    assert _rel(g1, r1) <= max(tol, 2e-5) and _rel(g2, r2) <= max(tol, 2e-5), (_rel(g1, r1), _rel(g2, r2))


def test_small_tail_routes_ties_to_the_first_maximum_and_is_reproducible():
    from tpgan_amd import ops
    torch.manual_seed(3)
    K, P = 12, 640
    base = torch.randn(P, 1, 16, device="cuda")
    h = base.expand(P, K, 16).reshape(P * K, 16).contiguous()            # K identical edges per point: every channel ties
    W1 = torch.randn(16, 16, device="cuda", requires_grad=True)
    W2 = torch.randn(32, 16, device="cuda", requires_grad=True)
    hi = h.clone().requires_grad_(True)
    y = ops.small_tail(hi, W1, W2, 0.2, 0.2, K)
    gh, g1, g2 = torch.autograd.grad(y.sum(), [hi, W1, W2])
    gh = gh.view(P, K, 16)
    assert float(gh[:, 1:].abs().max()) == 0.0 and float(gh[:, 0].abs().max()) > 0.0    # all of it on edge 0
    hi2 = h.clone().requires_grad_(True)
    gh2, g1b, g2b = torch.autograd.grad(ops.small_tail(hi2, W1, W2, 0.2, 0.2, K).sum(), [hi2, W1, W2])
    assert torch.equal(gh.reshape(-1, 16), gh2) and torch.equal(g1, g1b) and torch.equal(g2, g2b)


@pytest.mark.parametrize("amp", [False, True])
def test_idgcn_small_tails_equal_the_separate_launches(amp):
    """IDGCNLayer.forward_rows (two EdgeConvs with 16 -> 16 -> 32 tails) with the one-launch tails against the
    GEMM + activation launches they replace.  fp32: the two forms sum in different orders, outputs agree to 1e-4;
    a near-tie of the max over the neighbours may then pick another edge and move its share of a gradient
    (measured 6e-4 of a weight gradient's norm): 2e-3.  bf16 autocast: the unfused path rounds every
    intermediate to bf16, the fused one only its output (measured 3.1e-2 on one weight gradient): 5e-2."""
    from tpgan_amd import graph_conv
    from tpgan_amd.graph_conv import IDGCNLayer
    torch.manual_seed(6)
    m = IDGCNLayer(128, 128, bn=False, insn=False, residual=True).cuda()
    x = torch.randn(4, 512, 128, device="cuda")
    res = []
    for fused in (True, False):
        graph_conv.FUSED_EDGE_TAILS[0] = fused
        try:
            xi = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                y = m.forward_rows(xi)
            g = torch.autograd.grad(y.float().square().sum(), [xi] + list(m.parameters()))
            res.append((y.float(),) + tuple(t.float() for t in g))
        finally:
            graph_conv.FUSED_EDGE_TAILS[0] = True
    assert _rel(res[0][0], res[1][0]) <= (3e-2 if amp else 1e-4)
    for a, b in zip(*res):
        assert a.shape == b.shape and _rel(a, b) <= (5e-2 if amp else 2e-3), _rel(a, b)


# ------------------------------------------------------------------ row-wise linear layers (csrc/rowlinear.hip)
@pytest.mark.parametrize("P,Cin,Cout,nseg", [
    (12288, 3, 64, 1), (12288, 128, 128, 1), (12288, 128, 32, 1), (12288, 32, 16, 1), (12288, 96, 128, 1),
    (12288, 256, 64, 1), (12288, 256, 12, 1), (12288, 12, 24, 1), (12288, 24, 24, 1), (12288, 64, 1, 1),
    (65536, 6, 64, 2), (6144, 131, 128, 6), (4096, 259, 256, 2), (3072, 515, 256, 4), (8, 256, 256, 1),
    (8, 64, 1, 1), (1, 5, 7, 1), (33, 17, 19, 1), (100, 300, 130, 1), (256, 16, 16, 2), (4096, 256, 256, 1), (2048, 515, 256, 1)])
@pytest.mark.parametrize("din,dout", [("f32", "f32"), ("bf16", "bf16"), ("f32", "bf16")])
def test_row_linear_against_pytorch_fp32(P, Cin, Cout, nseg, din, dout):
    """tpg_rowlinear_fwd / dgrad / wgrad against plain PyTorch fp32 on the same (rounded) rows: the generator's and
    the discriminators' channel counts (3, 6, 12, 24, 96, 131, 259, 515, ...), segments with their own weights,
    bias and LeakyReLU in the epilogue, tiny P (the heads), ragged sizes.  fp32 in / out: 2e-5 of the output scale
    (fp32 products, another summation order); bf16: one rounding of the stored values."""
    import tpgan_amd.ops as ops
    dt = {"f32": torch.float32, "bf16": torch.bfloat16}
    torch.manual_seed(P + Cin + Cout)
    dev = "cuda"
    assert ops.ROW_LINEAR[0] is False           # (off by default: slower than the library, see ops.ROW_LINEAR; the op itself is tested)
    x = torch.randn(P, Cin, device=dev).to(dt[din])
    W = (torch.randn(nseg, Cout, Cin, device=dev) / Cin ** 0.5)
    b = torch.randn(Cout, device=dev) * 0.3
    gy = torch.randn(P, Cout, device=dev).to(dt[dout])
    for slope, use_b in ((0.2, True), (1.0, False)):
        xa = x.clone().requires_grad_(True)
        Wa = (W if nseg > 1 else W[0]).clone().requires_grad_(True)
        ba = b.clone().requires_grad_(True) if use_b else None
        y = ops.row_linear(xa, Wa, ba, slope, nseg, dt[dout])
        assert y.dtype == dt[dout] and y.shape == (P, Cout)
        y.backward(gy)
        # reference: fp32 math on the same inputs
        xr = x.float().clone().requires_grad_(True)
        Wr = W.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True) if use_b else None
        z = torch.bmm(xr.view(nseg, P // nseg, Cin), Wr.transpose(1, 2)).view(P, Cout)
        if use_b:
            z = z + br
        yr = torch.nn.functional.leaky_relu(z, slope) if slope != 1.0 else z
        # the kernel's backward takes lrelu' from the sign of the STORED output
        ys = y.detach().float()
        gz = gy.float() * torch.where(ys > 0, torch.ones_like(ys), torch.full_like(ys, slope)) if slope != 1.0 else gy.float()
        z.backward(gz)
        tol_y = 2e-5 if dout == "f32" else 8e-3
        assert float((y.float() - yr).abs().max()) <= tol_y * max(1.0, float(yr.abs().max())), (slope, float((y.float() - yr).abs().max()))
        tol_g = 2e-5 if din == "f32" else 8e-3
        assert float((xa.grad.float() - xr.grad).abs().max()) <= tol_g * max(1.0, float(xr.grad.abs().max()))
        gW = Wa.grad.view(nseg, Cout, Cin)
        assert float((gW - Wr.grad).abs().max()) <= 1e-4 * max(1.0, float(Wr.grad.abs().max())), float((gW - Wr.grad).abs().max())
        if use_b:
            assert float((ba.grad - br.grad).abs().max()) <= 1e-4 * max(1.0, float(br.grad.abs().max()))


def test_row_linear_declines_what_it_cannot_stage():
    """Channel counts whose 16-row chunk does not fit the weight-gradient kernel's LDS are reported unsupported (the
    host layer then keeps the library GEMM) instead of failing inside a launch."""
    import tpgan_amd.ops as ops
    x = torch.randn(256, 515, device="cuda")
    prev, ops.ROW_LINEAR[0] = ops.ROW_LINEAR[0], True
    try:
        assert ops.row_linear_supported(x, torch.randn(256, 515, device="cuda"))
        assert not ops.row_linear_supported(x, torch.randn(512, 515, device="cuda"))
        assert not ops.row_linear_supported(torch.randn(256, 2000, device="cuda"), torch.randn(64, 2000, device="cuda"))
    finally:
        ops.ROW_LINEAR[0] = prev


def test_row_linear_is_bitwise_reproducible():
    import tpgan_amd.ops as ops
    torch.manual_seed(1)
    x = torch.randn(12288, 96, device="cuda")
    W = torch.randn(128, 96, device="cuda").requires_grad_(True)
    b = torch.randn(128, device="cuda").requires_grad_(True)
    gy = torch.randn(12288, 128, device="cuda")
    res = []
    for _ in range(2):
        xa = x.clone().requires_grad_(True)
        y = ops.row_linear(xa, W, b, 0.2)
        res.append((y.detach().clone(),) + tuple(g.clone() for g in torch.autograd.grad(y, [xa, W, b], gy)))
    for a, c in zip(*res):
        assert torch.equal(a, c)
