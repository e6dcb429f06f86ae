"""Parity tests proper: every HIP entry point, called through the C-ABI, against the
oracle on the same seeded inputs.  Neighbour / sample indices must be BIT-EXACT, distances
produced by the canonical no-FMA sum must be bit-exact too; scatter-add gradients (LDS /
global float atomics, order not fixed) are held to 1e-5."""
import copy

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
TOL = 1e-5  # fp32 feature / gradient tolerance stated by BASELINE.json north_star


@pytest.fixture(scope="module")
def hip():
    import tpgan_amd.ops as ops
    assert torch.cuda.is_available()
    return ops.backend_for(torch.zeros(1, device="cuda"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def fluid(rng, B, N, scale=None):
    """Uniform box with the particle density of the synthetic fluid clips (spacing 0.025):
    half-width 0.25 at N = 4096, growing with N^(1/3) beyond it."""
    if scale is None:
        scale = 0.25 * max(1.0, (N / 4096.0) ** (1.0 / 3.0))
    return rng.uniform(-scale, scale, (B, N, 3)).astype(np.float32)


# ------------------------------------------------------------------ kNN
@pytest.mark.parametrize("B,P1,P2,D,K", [
    (2, 512, 512, 3, 20), (2, 512, 512, 32, 9), (2, 512, 512, 32, 20), (2, 512, 512, 64, 12),
    (2, 512, 512, 64, 4), (2, 512, 512, 64, 8), (3, 256, 256, 3, 32), (1, 100, 333, 5, 7),
    (1, 70, 129, 16, 64), (2, 33, 4096, 3, 16), (1, 5, 3, 3, 8), (1, 64, 64, 3, 1),
    (1, 16384, 16384, 3, 16), (1, 2048, 16384, 3, 32)])     # cfg5 cloud size
def test_knn_bit_exact(hip, B, P1, P2, D, K):
    rng = np.random.default_rng(B * 1000 + P1 + D + K)
    p1 = rng.standard_normal((B, P1, D)).astype(np.float32)
    p2 = p1.copy() if P1 == P2 else rng.standard_normal((B, P2, D)).astype(np.float32)
    if P2 > 20:
        p2[0, 5] = p2[0, 7]; p2[-1, 10:14] = p2[-1, 3]   # exact duplicates
    d, i = hip.knn(dev(p1), dev(p2), None, None, K, None)
    rd, ri = R.knn(p1, p2, K)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)


def _feature_cloud(rng, B, P, D, kind):
    """Feature-space clouds as the generator's bottleneck layers produce them: a few-dimensional manifold
    embedded in D dims plus noise ("manifold"), iid normal ("normal"), the same far from the origin
    ("offset": the Gram form cancels badly unless centred), and a cloud of 1/4 exact duplicates ("dups")."""
    if kind == "manifold":
        z = rng.standard_normal((B, P, 3)).astype(np.float32)
        x = np.tanh(z @ rng.standard_normal((3, D)).astype(np.float32)) + 0.05 * rng.standard_normal((B, P, D)).astype(np.float32)
    else:
        x = rng.standard_normal((B, P, D)).astype(np.float32)
    if kind == "lattice":
        # small integers: distances are small integers too -- exact ties everywhere, in particular AT the K-th
        # distance, where only the (dist, idx) order decides and the filter's margin test must give up
        x = rng.integers(0, 3, (B, P, D)).astype(np.float32) * 0.5
    if kind == "offset":
        x += 300.0
    if kind == "dups":
        x[:, P // 4 * 3:] = x[:, :P - P // 4 * 3]
    return np.ascontiguousarray(x.astype(np.float32))


@pytest.mark.parametrize("B,P1,P2,D,K,kind", [
    (2, 4096, 4096, 32, 20, "manifold"), (1, 4096, 4096, 64, 12, "manifold"), (1, 2048, 2048, 32, 10, "normal"),
    (1, 3000, 5000, 64, 8, "normal"), (1, 4096, 4096, 32, 20, "offset"), (1, 2500, 2500, 32, 20, "dups"),
    (1, 700, 16384, 64, 24, "manifold"), (1, 2048, 2048, 32, 20, "lattice"), (1, 600, 4096, 64, 12, "lattice")])
def test_knn_matrix_core_filter_bit_exact(hip, B, P1, P2, D, K, kind):
    """tpg_knn_f32 on clouds of >= 2048 points in 32 / 64 dims runs the Gram filter on the f32 matrix cores and
    re-ranks the survivors exactly (csrc/knn_mfma.hpp): indices and distances equal the oracle's bit for bit,
    whatever share of the queries fell back to the exhaustive kernel -- and on benign clouds that share is small
    (a wrong operand layout would send every query there and still pass the first assertion)."""
    rng = np.random.default_rng(P1 + P2 + D + K)
    p2 = _feature_cloud(rng, B, P2, D, kind)
    p1 = p2 if P1 == P2 else _feature_cloud(rng, B, P1, D, kind)
    d, i = hip.knn(dev(p1), dev(p2), None, None, K, None)
    rd, ri = R.knn(p1, p2, K)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)
    _, raw = hip.knn_mfma(dev(p1), dev(p2), None, None, K, redo=False)
    raw = raw.cpu().numpy()
    open_ = raw[:, :, 0] == -2
    print(f"matrix-core kNN {kind} D={D} K={K} P2={P2}: {open_.mean() * 100:.2f} % of the queries left to the exhaustive kernel")
    assert np.array_equal(raw[~open_], ri[~open_])           # what the filter settled IS the answer
    if kind in ("manifold", "normal"):
        assert open_.mean() < 0.02


def test_knn_matrix_core_filter_ragged(hip):
    rng = np.random.default_rng(77)
    p1, p2 = _feature_cloud(rng, 3, 2300, 32, "normal"), _feature_cloud(rng, 3, 4100, 32, "normal")
    l1, l2 = np.array([2300, 1000, 0]), np.array([4100, 2049, 40])
    d, i = hip.knn(dev(p1), dev(p2), dev(l1), dev(l2), 16, None)
    rd, ri = R.knn(p1, p2, 16, l1, l2)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)


def test_knn_all_identical_points(hip):
    p = np.full((1, 200, 3), 999.0, np.float32)   # a cloud of hard-masking dummies
    d, i = hip.knn(dev(p), dev(p), None, None, 16, None)
    rd, ri = R.knn(p, p, 16)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)


def test_knn_sorted_descending_worst_case(hip):
    # candidates arrive in strictly decreasing distance: every candidate is inserted
    x = np.zeros((1, 300, 3), np.float32)
    x[0, :, 0] = np.linspace(30, 0.1, 300)
    q = np.zeros((1, 4, 3), np.float32)
    d, i = hip.knn(dev(q), dev(x), None, None, 20, None)
    rd, ri = R.knn(q, x, 20)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)


def test_knn_ragged(hip):
    rng = np.random.default_rng(4)
    p1, p2 = fluid(rng, 3, 40), fluid(rng, 3, 130)
    l1, l2 = np.array([40, 7, 0]), np.array([130, 4, 65])
    d, i = hip.knn(dev(p1), dev(p2), dev(l1), dev(l2), 6, None)
    rd, ri = R.knn(p1, p2, 6, l1, l2)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("B,P1,P2,K,r", [
    (2, 512, 4096, 1, 0.0475), (2, 4096, 4096, 16, 0.035), (3, 256, 256, 32, 2.0),
    (1, 100, 300, 8, 0.01), (1, 64, 64, 64, 0.2),
    (1, 4096, 16384, 1, 0.0475), (1, 16384, 16384, 16, 0.035)])   # cfg5: masking_loss searches (loss.py:256-265)
def test_frnn_bit_exact(hip, B, P1, P2, K, r):
    import tpgan_amd.ops as ops
    rng = np.random.default_rng(K)
    p2 = fluid(rng, B, P2)
    p1 = p2.copy() if P1 == P2 else fluid(rng, B, P1)
    d, i = hip.knn(dev(p1), dev(p2), None, None, K, ops.radius_sq(r))
    rd, ri = R.knn(p1, p2, K, r=r)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)
    if r <= 0.035:
        assert (ri == -1).any()   # the small radii really exercise the -1 padding


# ------------------------------------------------------------------ Chamfer
@pytest.mark.parametrize("B,N,M", [(2, 4096, 4096), (3, 500, 777), (1, 1, 1), (2, 65, 64)])
def test_chamfer_fwd_and_bwd_bit_exact(hip, B, N, M):
    rng = np.random.default_rng(N + M)
    s, t = fluid(rng, B, N), fluid(rng, B, M)
    d1, i1, d2, i2 = hip.chamfer_fwd(dev(s), dev(t))
    r1, ri1, r2, ri2 = R.chamfer_fwd(s, t)
    assert np.array_equal(i1.cpu().numpy(), ri1) and np.array_equal(i2.cpu().numpy(), ri2)
    assert np.array_equal(d1.cpu().numpy(), r1) and np.array_equal(d2.cpu().numpy(), r2)
    g1 = rng.standard_normal((B, N)).astype(np.float32)
    g2 = rng.standard_normal((B, M)).astype(np.float32)
    gs, gt = hip.chamfer_bwd(dev(s), dev(t), i1, i2, dev(g1), dev(g2))
    rs, rt = R.chamfer_bwd(s, t, ri1, ri2, g1, g2)
    # round 3: no float atomics -- every point sums its own term and the terms of the points that chose it in the
    # oracle's loop order (tpgref_chamfer_bwd_f32), so the gradients are the oracle's bit for bit
    assert np.array_equal(gs.cpu().numpy(), rs)
    assert np.array_equal(gt.cpu().numpy(), rt)


def test_chamfer_bwd_many_points_on_one_neighbour(hip):
    """A target cloud collapsed near one source point: its inverted list holds every target point (the float
    atomics of round 2 summed those in arrival order); and the same launch twice gives the same bits."""
    rng = np.random.default_rng(8)
    B, N, M = 2, 700, 3000
    s = fluid(rng, B, N)
    t = (s[:, 5:6, :] + 1e-3 * rng.standard_normal((B, M, 3))).astype(np.float32)
    d1, i1, d2, i2 = hip.chamfer_fwd(dev(s), dev(t))
    assert (np.bincount(i2[0].cpu().numpy(), minlength=N).max() > M // 2)
    g1 = rng.standard_normal((B, N)).astype(np.float32)
    g2 = rng.standard_normal((B, M)).astype(np.float32)
    gs, gt = hip.chamfer_bwd(dev(s), dev(t), i1, i2, dev(g1), dev(g2))
    rs, rt = R.chamfer_bwd(s, t, i1.cpu().numpy(), i2.cpu().numpy(), g1, g2)
    assert np.array_equal(gs.cpu().numpy(), rs) and np.array_equal(gt.cpu().numpy(), rt)
    gs2, gt2 = hip.chamfer_bwd(dev(s), dev(t), i1, i2, dev(g1), dev(g2))
    assert torch.equal(gs, gs2) and torch.equal(gt, gt2)


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,m", [
    (2, 4096, 1024), (3, 1024, 512), (2, 1024, 256), (2, 512, 128), (2, 200, 64), (1, 64, 64),
    (2, 2048, 512), (1, 5000, 300), (1, 9000, 200), (1, 20000, 64), (1, 10, 25), (4, 1, 3),
    (2, 16384, 4096)])                                       # cfg5 first level
def test_fps_bit_exact(hip, B, N, m):
    rng = np.random.default_rng(N + m)
    x = fluid(rng, B, N)
    if N > 8:
        x[0, 3] = 0.0                 # origin point: skipped
        x[-1, N // 2: N // 2 + 3] = 999.0   # dummies with identical coordinates
    idx = hip.fps(dev(x), m)
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy(), R.fps(x, m))


@pytest.mark.parametrize("case", ["clusters", "line", "one_point", "few_distinct", "none_eligible", "ragged_8192",
                                  "ragged_12289", "smallest_pruned", "start_all_eligible", "far_coordinates", "clusters_16384", "few_distinct_9000"])
def test_fps_pruned_rounds_bit_exact(hip, case):
    """The pruned rounds of csrc/fps.hip (Morton-ordered tiles, box bounds, tile records; the kernel takes clouds of
    8193..16384 points and >= 128 picks -- the smaller cases here keep the dense kernel honest on the same shapes)
    against the oracle, bit for bit, on the shapes that stress the pruning: far-apart clusters (most
    tiles skipped from the first rounds on), degenerate extents, ties everywhere, no eligible point, clouds that do not
    fill the last tiles, picks by original index when tiles hold points out of index order."""
    rng = np.random.default_rng(len(case))
    start, skip = None, True
    if case == "clusters":
        x = np.concatenate([fluid(rng, 2, 2048) * 0.05 + 0.4, fluid(rng, 2, 2048) * 0.05 - 0.4], axis=1)
        x = x[:, rng.permutation(4096)]
        m = 1024
    elif case == "line":
        x = np.zeros((2, 9000, 3), np.float32)
        x[..., 0] = rng.uniform(0.1, 1.0, (2, 9000))
        x[..., 1] = 0.3
        m = 512
    elif case == "one_point":
        x = np.full((1, 9000, 3), 0.25, np.float32)
        m = 200
    elif case == "few_distinct":
        few = fluid(rng, 1, 40)[0]
        x = few[rng.integers(0, 40, (2, 5000))]
        m = 300
    elif case == "none_eligible":
        x = np.full((2, 10000, 3), 0.01, np.float32)
        x[0] = fluid(rng, 1, 10000)[0]
        m = 256
    elif case == "ragged_8192":
        x = fluid(rng, 2, 8190)
        m = 2048
    elif case == "ragged_12289":
        x = fluid(rng, 1, 12289)
        x[0, 5] = 0.0
        m = 700
    elif case == "smallest_pruned":
        x = fluid(rng, 3, 2049)
        m = 128
    elif case == "start_all_eligible":
        x = (rng.standard_normal((2, 9216, 3)) * 0.3).astype(np.float32)
        x[0, 7] = 0.0
        start, skip, m = rng.integers(0, 9216, 2).astype(np.int32), False, 1152
    elif case == "clusters_16384":
        x = np.concatenate([fluid(rng, 1, 8192) * 0.05 + 0.4, fluid(rng, 1, 8192) * 0.05 - 0.4], axis=1)
        x = x[:, rng.permutation(16384)]
        m = 2048
    elif case == "few_distinct_9000":
        few = fluid(rng, 1, 40)[0]
        x = few[rng.integers(0, 40, (2, 9000))]
        m = 300
    else:
        x = fluid(rng, 2, 9000) + np.float32(2.0e5)          # box bounds beyond the 1e10 start value of the distances
        m = 300
    x = np.ascontiguousarray(x, np.float32)
    if start is None:
        got = hip.fps(dev(x), m).cpu().numpy()
        want = R.fps(x, m)
    else:
        got = hip.fps(dev(x), m, dev(start), skip).cpu().numpy()
        want = R.fps_start(x, m, start, skip_origin=skip)
    assert np.array_equal(got, want), (case, np.argwhere(got != want)[:5])


def test_fps_all_points_inside_origin_ball(hip):
    x = np.full((2, 300, 3), 0.01, np.float32)
    assert np.array_equal(hip.fps(dev(x), 16).cpu().numpy(), R.fps(x, 16))


def test_fps_duplicate_points_tie_to_smallest_index(hip):
    rng = np.random.default_rng(0)
    x = fluid(rng, 1, 128)
    x = np.concatenate([x, x], axis=1)   # every point twice (MSR-style repeats)
    assert np.array_equal(hip.fps(dev(x), 100).cpu().numpy(), R.fps(x, 100))


def _gather_rows(x, idx):
    return np.take_along_axis(x, idx[..., None].astype(np.int64), axis=1)


@pytest.mark.parametrize("B,N,chain", [(3, 4096, (1024, 512, 128)), (4, 4096, (1024, 256)), (2, 16384, (1024, 512)),
                                       (1, 20000, (300, 100)), (2, 2048, (512, 256, 128)), (2, 300, (64, 64, 7))])
def test_fps_of_an_fps_prefix_is_the_full_algorithm(hip, B, N, chain):
    """tpg_fps_prefix_f32 (round 3): a set-abstraction level samples the centres of the level before, in pick order,
    and the sampling of such a prefix is 0..m-1 whenever the producing sampling's last maximum was positive.  Every
    level of a chain, with the flags handed on, against the FULL kernel and the oracle on the same gathered centres
    -- bit for bit -- and the flags must be set on benign clouds (so the shortcut really ran)."""
    rng = np.random.default_rng(N + len(chain))
    x = fluid(rng, B, N)
    x[:, N // 3] = x[:, N // 3 + 1]                      # a duplicate pair somewhere in the cloud
    cur, flag = x, None
    for level, m in enumerate(chain):
        idx, flag = hip.fps_prefix(dev(cur), m, flag)
        full = hip.fps(dev(cur), m)
        assert torch.equal(idx, full), (level, m)
        assert np.array_equal(idx.cpu().numpy(), R.fps(cur, m)), (level, m)
        assert flag.cpu().tolist() == [1] * B, (level, flag.cpu().tolist())
        if level:
            assert np.array_equal(idx.cpu().numpy(), np.broadcast_to(np.arange(m, dtype=np.int32), (B, m)))
        cur = _gather_rows(cur, idx.cpu().numpy())


def test_fps_prefix_declines_where_the_maximum_reached_zero(hip):
    """Clouds with fewer distinct eligible points than picks: the producing sampling's last maximum is 0, its flag stays
    clear and the consuming launch runs the full algorithm (whose ties reach back to chosen positions: NOT 0..m-1) --
    per cloud, next to benign clouds of the same batch that do take the shortcut."""
    rng = np.random.default_rng(12)
    B, N, m0, m1 = 4, 1024, 256, 128
    x = fluid(rng, B, N)
    few = fluid(rng, 1, 40)[0]
    x[1] = few[rng.integers(0, 40, N)]                    # 40 distinct points, 1024 copies
    x[3, :, :] = 0.01                                     # every point inside the origin ball: none eligible
    idx0, flag0 = hip.fps_prefix(dev(x), m0, None)
    assert np.array_equal(idx0.cpu().numpy(), R.fps(x, m0))
    assert flag0.cpu().tolist() == [1, 0, 1, 0]
    x1 = _gather_rows(x, idx0.cpu().numpy())
    idx1, flag1 = hip.fps_prefix(dev(x1), m1, flag0)
    want = R.fps(x1, m1)
    assert np.array_equal(idx1.cpu().numpy(), want)
    assert np.array_equal(want[0], np.arange(m1)) and not np.array_equal(want[1], np.arange(m1))
    assert flag1.cpu().tolist() == [1, 0, 1, 0]
    # action-clip duplicates (1/8 of the points repeated) with enough distinct points: the shortcut holds
    y = (0.4 * rng.standard_normal((2, 2048, 3))).astype(np.float32)
    y[:, -256:] = y[:, :256]
    i0, f0 = hip.fps_prefix(dev(y), 512, None)
    y1 = _gather_rows(y, i0.cpu().numpy())
    i1, f1 = hip.fps_prefix(dev(y1), 256, f0)
    assert f0.cpu().tolist() == [1, 1] and np.array_equal(i1.cpu().numpy(), R.fps(y1, 256))


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize("B,N,S,r,ns", [
    (2, 4096, 1024, 0.10, 32), (2, 4096, 1024, 0.15, 32), (2, 1024, 512, 0.30, 32),
    (2, 512, 128, 0.60, 16), (2, 1024, 256, 0.20, 32), (1, 2048, 512, 0.8, 64),
    (1, 9000, 77, 0.05, 16), (1, 100, 3, 0.001, 8), (3, 70, 70, 0.2, 100),
    (2, 16384, 1024, 0.10, 32), (1, 16384, 4096, 0.05, 32)])    # cfg5 first level
def test_ball_query_bit_exact(hip, B, N, S, r, ns):
    rng = np.random.default_rng(N + S + ns)
    x = fluid(rng, B, N)
    q = x[:, rng.permutation(N)[:S]].copy()
    if S > 2:
        q[0, 1] = 50.0    # a query with no neighbour at all -> zero row
    idx = hip.ball_query(r, ns, dev(x), dev(q))
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy(), R.ball_query(r, ns, x, q))


# ------------------------------------------------------------------ group / gather
@pytest.mark.parametrize("B,C,N,S,K", [
    (2, 3, 512, 512, 20), (2, 32, 512, 512, 9), (2, 64, 512, 512, 12), (2, 128, 1024, 512, 32),
    (2, 256, 256, 256, 32), (2, 3, 4096, 1024, 32), (1, 5, 100, 33, 7), (1, 1, 1, 1, 1),
    (1, 9, 6000, 50, 3), (1, 2, 20000, 10, 4)])
def test_group_fwd_exact_bwd_close(hip, B, C, N, S, K):
    rng = np.random.default_rng(C + N + S + K)
    f = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, S, K)).astype(np.int32)
    idx[0, 0, :] = 0   # heavy repetition on one row
    out = hip.group_fwd(dev(f), dev(idx))
    assert np.array_equal(out.cpu().numpy(), R.group_fwd(f, idx))
    g = rng.standard_normal((B, C, S, K)).astype(np.float32)
    gf = hip.group_bwd(dev(g), dev(idx), N).cpu().numpy()
    ref = R.group_bwd(g, idx, N)
    assert np.abs(gf - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_group_bwd_is_deterministic_for_permutation_indices(hip):
    # every source index used exactly once -> no accumulation order -> bit-exact
    rng = np.random.default_rng(1)
    N = 512
    idx = np.stack([rng.permutation(N) for _ in range(2)]).astype(np.int32).reshape(2, N // 8, 8)
    g = rng.standard_normal((2, 16, N // 8, 8)).astype(np.float32)
    assert np.array_equal(hip.group_bwd(dev(g), dev(idx), N).cpu().numpy(), R.group_bwd(g, idx, N))


@pytest.mark.parametrize("B,C,N,S", [(2, 3, 4096, 1024), (2, 3, 1024, 256), (1, 7, 50, 300), (1, 1, 1, 1)])
def test_gather_fwd_exact_bwd_close(hip, B, C, N, S):
    rng = np.random.default_rng(N + S)
    f = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, S)).astype(np.int32)
    assert np.array_equal(hip.gather_fwd(dev(f), dev(idx)).cpu().numpy(), R.gather_fwd(f, idx))
    g = rng.standard_normal((B, C, S)).astype(np.float32)
    ref = R.gather_bwd(g, idx, N)
    got = hip.gather_bwd(dev(g), dev(idx), N).cpu().numpy()
    assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_out_of_range_indices_never_fault(hip):
    f = torch.arange(10, dtype=torch.float32, device="cuda").view(1, 1, 10).contiguous()
    idx = torch.tensor([[[-5, 3, 10, 1 << 30]]], dtype=torch.int32, device="cuda")
    out = hip.group_fwd(f, idx)
    assert out.view(-1).tolist() == [9.0, 3.0, 9.0, 9.0]   # clamped, documented in DESIGN.md


# ------------------------------------------------------------------ three_nn / interpolate
def test_three_nn_and_interpolate(hip):
    rng = np.random.default_rng(6)
    u, k = fluid(rng, 2, 300), fluid(rng, 2, 500)
    d2, idx = hip.three_nn(dev(u), dev(k))
    rd, ri = R.three_nn(u, k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(d2.cpu().numpy(), rd)
    f = rng.standard_normal((2, 6, 500)).astype(np.float32)
    w = rng.uniform(0, 1, (2, 300, 3)).astype(np.float32)
    out = hip.three_interp_fwd(dev(f), idx, dev(w)).cpu().numpy()
    assert np.array_equal(out, R.three_interp_fwd(f, ri, w))
    g = rng.standard_normal(out.shape).astype(np.float32)
    ref = R.three_interp_bwd(g, ri, w, 500)
    got = hip.three_interp_bwd(dev(g), idx, dev(w), 500).cpu().numpy()
    assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------ full-size properties
def test_full_size_properties_cfg2(hip):
    """BASELINE config 2 sizes (B=8, N_hi=4096): size-independent properties, no oracle."""
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.rand(8, 4096, 3, generator=g) - 0.5) * 0.5
    xd = x.cuda()
    # kNN: self is neighbour 0 at distance 0, rows sorted ascending, indices unique per row
    d, i = hip.knn(xd, xd, None, None, 16, None)
    assert torch.equal(i[:, :, 0], torch.arange(4096, device="cuda").expand(8, -1))
    assert (d[:, :, 0] == 0).all() and (d[:, :, 1:] >= d[:, :, :-1]).all()
    assert (torch.sort(i, dim=2)[0][:, :, 1:] != torch.sort(i, dim=2)[0][:, :, :-1]).all()
    # FRNN == kNN wherever the k-th neighbour lies inside the radius
    import tpgan_amd.ops as ops
    fd, fi = hip.knn(xd, xd, None, None, 16, ops.radius_sq(0.035))
    inside = d < ops.radius_sq(0.035)
    assert torch.equal(fi[inside], i[inside]) and (fi[~inside] == -1).all()
    # FPS: indices distinct, first is 0
    s = hip.fps(xd, 1024)
    assert (s[:, 0] == 0).all()
    assert all(len(set(row.tolist())) == 1024 for row in s.cpu())
    # ball query rows: sorted ascending until the fill, all within radius
    new = torch.gather(xd, 1, s.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    bq = hip.ball_query(0.15, 32, xd, new)
    nb = torch.gather(xd.unsqueeze(1).expand(-1, 1024, -1, -1), 2,
                      bq.long().unsqueeze(-1).expand(-1, -1, -1, 3))
    assert (((nb - new.unsqueeze(2)) ** 2).sum(-1) < 0.15 ** 2 + 1e-6).all()
    # group: linear in the features, and group(bwd(ones)) counts index multiplicities
    f = torch.randn(8, 128, 4096, device="cuda")
    o = hip.group_fwd(f, bq)
    assert torch.equal(o, torch.gather(f.unsqueeze(2).expand(-1, -1, 1024, -1), 3,
                                       bq.long().unsqueeze(1).expand(-1, 128, -1, -1)))
    cnt = hip.group_bwd(torch.ones(8, 1, 1024, 32, device="cuda"), bq, 4096)
    ref = torch.zeros(8, 4096, device="cuda").scatter_add_(1, bq.long().view(8, -1),
                                                           torch.ones(8, 1024 * 32, device="cuda"))
    assert torch.equal(cnt[:, 0], ref)
    # Chamfer of a cloud with itself is exactly 0 with identity assignment
    d1, i1, d2, i2 = hip.chamfer_fwd(xd, xd)
    assert (d1 == 0).all() and (d2 == 0).all()
    assert torch.equal(i1, torch.arange(4096, device="cuda").expand(8, -1))


@pytest.mark.parametrize("B,N,S,C", [(3, 4096, 1024, 3), (2, 512, 128, 3), (1, 100, 100, 6), (2, 64, 7, 1)])
def test_gather_rows_equals_gather_operation_on_the_transposed_tensors(hip, B, N, S, C):
    """tpg_gather_rows_*: bit-exact against the oracle (a pure copy; sums of the backward meet only on repeated centres)
    and against tpg_gather_*_f32 on the (B,C,N) planes."""
    from tpgan_amd import ops
    rng = np.random.default_rng(N + S)
    rows = rng.standard_normal((B, N, C)).astype(np.float32)
    idx = np.stack([rng.permutation(N)[:S] for _ in range(B)]).astype(np.int32)
    idx[0, :3] = idx[0, 3]                       # a centre picked several times
    x = dev(rows).requires_grad_(True)
    out = ops.gather_rows(x, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), R.gather_rows_fwd(rows, idx))
    g = rng.standard_normal((B, S, C)).astype(np.float32)
    (gx,) = torch.autograd.grad(out, x, dev(g))
    ref = R.gather_rows_bwd(g, idx, N)
    assert np.abs(gx.cpu().numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    x2 = dev(rows).requires_grad_(True)
    out2 = ops.gather_operation(x2.transpose(1, 2).contiguous(), dev(idx)).transpose(1, 2)
    assert torch.equal(out, out2)
    (gx2,) = torch.autograd.grad(out2, x2, dev(g))
    assert torch.allclose(gx, gx2, atol=1e-6)


# ------------------------------------------------------------------ row combine (channels-last)
def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).bfloat16().float().numpy()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,N,S,K,C", [(2, 512, 512, 20, 64), (2, 1024, 512, 32, 128), (1, 100, 100, 7, 8),
                                       (3, 256, 256, 32, 256), (1, 4096, 1024, 32, 16)])
@pytest.mark.parametrize("din,dout", [("f32", "f32"), ("bf16", "bf16"), ("f32", "bf16")])
def test_rowcombine_fwd_exact_bwd_close(hip, mode, B, N, S, K, C, din, dout):
    if mode == 2:
        S = N
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16}
    rng = np.random.default_rng(mode * 7 + C + K)
    U = rng.standard_normal((B, N, C)).astype(np.float32)
    Q = rng.standard_normal((B, S, C)).astype(np.float32)
    idx = rng.integers(0, N, (B, S, K)).astype(np.int32)
    idx[0, 0, :] = 3                      # one heavily repeated destination
    idx[B - 1, :, 0] = 5                  # a hub: one source row in EVERY group (a list of >= S entries)
    if din == "bf16":
        U, Q = _bf16_round(U), _bf16_round(Q)
    Ud, Qd = dev(U).to(tdt[din]), dev(Q).to(tdt[din])
    out = hip.rowcombine_fwd(Ud, None if mode == 0 else Qd, dev(idx), mode, 0.2, tdt[dout])
    ref = R.rowcombine_fwd(U, None if mode == 0 else Q, idx, mode, 0.2)
    if dout == "bf16":
        ref = _bf16_round(ref)
    assert out.dtype == tdt[dout]
    assert np.array_equal(out.float().cpu().numpy(), ref)          # one rounding, same arithmetic
    g = rng.standard_normal((B, S, K, C)).astype(np.float32)
    if dout == "bf16":
        g = _bf16_round(g)
    gU, gQ = hip.rowcombine_bwd(dev(g).to(tdt[dout]), dev(idx), Qd if mode == 2 else None, mode, N, 0.2,
                                tdt[din])
    rU, rQ = R.rowcombine_bwd(g, idx, Q if mode == 2 else None, mode, N, 0.2)
    tol = TOL if din == "f32" else 1e-2   # bf16 gradients are rounded to 8 bits on store
    assert gU.dtype == tdt[din]
    assert np.abs(gU.float().cpu().numpy() - rU).max() <= tol * max(1.0, np.abs(rU).max())
    if mode:
        assert np.abs(gQ.float().cpu().numpy() - rQ).max() <= tol * max(1.0, np.abs(rQ).max())
    else:
        assert gQ is None


@pytest.mark.parametrize("B,N,K,C", [(2, 512, 20, 64), (3, 512, 10, 16), (1, 100, 7, 8), (2, 1024, 12, 128), (1, 4096, 4, 128)])
@pytest.mark.parametrize("din,dout", [("f32", "f32"), ("f32", "bf16"), ("bf16", "bf16")])
def test_rowcombine_edge_front_fwd_exact_bwd_close(hip, B, N, K, C, din, dout):
    """tpg_rowcombine_edge_fwd / _bwd (the EdgeConv front end on one product, Y = f [We; Wn]^T) against the oracle:
    forward bit-exact (same fp32 arithmetic, one rounding on store), backward to summation order; the shapes are the
    generator's EdgeConvs (H = 64 k = 20, H = 16 k = 20 / 10, H = 128 k = 12 / 8 / 4) and a ragged one."""
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16}
    rng = np.random.default_rng(K * 31 + C)
    Y = rng.standard_normal((B, N, 2 * C)).astype(np.float32)
    idx = rng.integers(0, N, (B, N, K)).astype(np.int32)
    idx[0, 0, :] = 3
    idx[B - 1, :, 0] = 5                  # a hub row
    if din == "bf16":
        Y = _bf16_round(Y)
    Yd = dev(Y).to(tdt[din])
    out = hip.rowcombine_edge_fwd(Yd, dev(idx), 0.2, 0.1, tdt[dout])
    ref = R.rowcombine_edge_fwd(Y, idx, 0.2, 0.1)
    if dout == "bf16":
        ref = _bf16_round(ref)
    assert out.dtype == tdt[dout] and out.shape == (B, N, K, C)
    assert np.array_equal(out.float().cpu().numpy(), ref)
    g = rng.standard_normal((B, N, K, C)).astype(np.float32)
    if dout == "bf16":
        g = _bf16_round(g)
    gY = hip.rowcombine_edge_bwd(dev(g).to(tdt[dout]), dev(idx), Yd, 0.2, 0.1)
    rY = R.rowcombine_edge_bwd(g, idx, Y, 0.2, 0.1)
    tol = TOL if din == "f32" else 1e-2
    assert gY.dtype == tdt[din] and gY.shape == (B, N, 2 * C)
    assert np.abs(gY.float().cpu().numpy() - rY).max() <= tol * max(1.0, np.abs(rY).max())
    # and bitwise reproducible (ordered sums over the radix-sorted inverted index)
    gY2 = hip.rowcombine_edge_bwd(dev(g).to(tdt[dout]), dev(idx), Yd, 0.2, 0.1)
    assert torch.equal(gY, gY2)


@pytest.mark.parametrize("cin,H,K,flat", [(3, 64, 20, False), (32, 16, 10, True), (64, 128, 12, True)])
def test_edge_front_function_equals_two_products(hip, cin, H, K, flat):
    """graph_conv.edge_front (one GEMM + tpg_rowcombine_edge_*) against the form it replaces -- two GEMMs, LeakyReLU,
    ops.row_combine(ROW_EDGE) -- outputs and all three gradients; `flat`: the two weights back to back in one buffer
    (the stacked weight is then a view, as inside the graphed step)."""
    from tpgan_amd import graph_conv, ops
    torch.manual_seed(cin + H)
    B, N = 2, 512
    x = torch.randn(B, N, cin, device="cuda", requires_grad=True)
    if flat:
        buf = torch.randn(2 * H * cin, device="cuda") * 0.3
        we, wn = buf[:H * cin].view(H, cin), buf[H * cin:].view(H, cin)
        assert graph_conv._stacked_weight(we, wn).data_ptr() == buf.data_ptr()
    else:
        we, wn = torch.randn(H, cin, device="cuda") * 0.3, torch.randn(H, cin, device="cuda") * 0.3
    we, wn = we.detach().requires_grad_(), wn.detach().requires_grad_()
    idx = torch.randint(0, N, (B, N, K), device="cuda", dtype=torch.int32)
    g = torch.randn(B, N, K, H, device="cuda")
    h = graph_conv.edge_front(x, we, wn, idx, 0.2, 0.2, torch.float32)
    gx, gwe, gwn = torch.autograd.grad(h, [x, we, wn], g)
    A = torch.nn.functional.leaky_relu(x @ wn.t(), 0.2)
    E = x @ we.t()
    h0 = ops.row_combine(A, E, idx, ops.ROW_EDGE, slope=0.2, out_dtype=torch.float32)
    rx, rwe, rwn = torch.autograd.grad(h0, [x, we, wn], g)
    assert (h - h0).abs().max() <= 1e-5 * max(1.0, h0.abs().max())
    for a, b in ((gx, rx), (gwe, rwe), (gwn, rwn)):
        assert (a - b).abs().max() <= 2e-5 * max(1.0, b.abs().max())


@pytest.mark.parametrize("N,SK", [(1000, 7777), (16384, 4096 * 32), (16352, 5000), (16353, 5000),
                                  (40928, 70001), (100000, 150000), (200, 999), (256, 64), (257, 1), (2, 5000),
                                  (1, 130), (4096, 81920), (512, 512 * 20), (32704, 3000), (32705, 3000),
                                  (70000, 20), (16384, 16384 * 20), (140000, 1200000)])
def test_invert_index_is_the_stable_sort_by_destination(hip, N, SK):
    """tpg_invert_index == a STABLE argsort of the destinations, bit for bit (round 3: the order inside a list
    is the order of the backward's float sums).  Sizes: cfg2 / cfg5 clouds, one / two / three radix passes
    (N <= 256, <= 65536, beyond), packed and unpacked intermediate words (N = 16384 with 327680 entries needs
    7 + 19 bits; 140000 destinations x 1.2 M entries need 12 + 21: the later passes gather idx[e] again), the
    boundary of the LDS counter path (32704 rows) and clouds beyond it (offs from the sorted list), waves
    without entries, a single destination."""
    rng = np.random.default_rng(2)
    B = 3
    idx = rng.integers(0, N, (B, SK)).astype(np.int32)
    idx[1, :] = min(5, N - 1)             # every entry on one destination
    idx[2, : SK // 2] = np.sort(idx[2, : SK // 2])[::-1]           # long descending runs
    idx[2, SK // 2:] = idx[2, SK // 2:] // 7 * 7 % N               # many empty destinations
    offs, lst = hip.invert_index(dev(idx).view(B, SK, 1), N)
    offs, lst = offs.cpu().numpy(), lst.cpu().numpy()
    for b in range(B):
        want = np.argsort(idx[b], kind="stable")
        assert np.array_equal(lst[b], want), (b, np.flatnonzero(lst[b] != want)[:5])
        cnt = np.bincount(idx[b], minlength=N)
        assert np.array_equal(offs[b], np.concatenate([[0], np.cumsum(cnt)]))


def test_invert_index_clamps_like_the_forward_reads(hip):
    """Out-of-range indices land where tpg_rowcombine_fwd reads them (tpg_clamp_idx: anything outside [0, N) -> N-1)."""
    N, SK = 300, 1000
    rng = np.random.default_rng(4)
    idx = rng.integers(-50, N + 50, (2, SK)).astype(np.int32)
    offs, lst = hip.invert_index(dev(idx).view(2, SK, 1), N)
    cl = np.where((idx >= 0) & (idx < N), idx, N - 1)
    for b in range(2):
        assert np.array_equal(lst[b].cpu().numpy(), np.argsort(cl[b], kind="stable"))
        assert np.array_equal(np.diff(offs[b].cpu().numpy()), np.bincount(cl[b], minlength=N))


def test_rowcombine_bwd_at_cfg5_cloud_size(hip):
    """Row gather backward into N = 16384 source rows (cfg5's clouds: 4 x 4096 predictions / the real
    clouds entering both discriminators' first level): index-sorted ball-query-like lists."""
    rng = np.random.default_rng(5)
    B, N, S, K, C = 2, 16384, 4096, 32, 64
    idx = np.sort(rng.integers(0, N, (B, S, K)), axis=-1).astype(np.int32)
    idx[:, :, 0] = rng.integers(0, 64, (B, S))          # hubs among the low indices, like a ball query
    g = _bf16_round(rng.standard_normal((B, S, K, C)).astype(np.float32))
    gU, gQ = hip.rowcombine_bwd(dev(g).bfloat16(), dev(idx), None, 1, N, 0.2, torch.float32)
    rU, rQ = R.rowcombine_bwd(g, idx, None, 1, N, 0.2)
    assert np.abs(gU.cpu().numpy() - rU).max() <= TOL * max(1.0, np.abs(rU).max())
    assert np.abs(gQ.cpu().numpy() - rQ).max() <= TOL * max(1.0, np.abs(rQ).max())


# ------------------------------------------------------------------ fused BN + act (+max) on rows
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("P,K,C", [(8 * 512 * 32, 32, 128), (4096, 0, 64), (2 * 128 * 16, 16, 256), (3000, 0, 24),
                                   (2 * 256, 256, 256), (4 * 1024 * 32, 0, 64)])
@pytest.mark.parametrize("din,dout", [("f32", "f32"), ("bf16", "bf16"), ("f32", "bf16")])
def test_rowbn_matches_oracle(hip, training, P, K, C, din, dout):
    if dout == "bf16" and C % 8:
        pytest.skip("bf16 rows need C % 8 == 0")
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16}
    rng = np.random.default_rng(P + K + C)
    x = (rng.standard_normal((P, C)) * 1.7 + rng.standard_normal(C) * 3.0).astype(np.float32)
    if din == "bf16":
        x = _bf16_round(x)
    gamma = rng.uniform(0.5, 1.5, C).astype(np.float32)
    gamma[::5] = 0.05                      # |beta| > 4|gamma| there: the backward's gather branch
    beta = rng.standard_normal(C).astype(np.float32)
    rm0, rv0 = rng.standard_normal(C).astype(np.float32), rng.uniform(0.5, 2.0, C).astype(np.float32)
    slope, eps, mom = 0.01, 1e-5, 0.1
    nbt = torch.full((), 7, dtype=torch.int64, device="cuda")
    mean, rstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rm, rv = dev(rm0.copy()), dev(rv0.copy())
    if not training:
        mean, rstd = dev(rm0.copy()), torch.rsqrt(dev(rv0.copy()) + eps)
    xd = dev(x).to(tdt[din])
    y, arg = hip.rowbn_fwd(xd, K, eps, mom, training, rm if training else None, rv if training else None,
                           dev(gamma), dev(beta), slope, mean, rstd, tdt[dout], nbt if training else None)
    assert int(nbt) == (8 if training else 7)          # incremented by the statistics launch
    ry, rmean, rrstd, rarg = R.rowbn_fwd(x, K, eps, gamma, beta, slope, training,
                                         None if training else rm0, None if training else 1 / np.sqrt(rv0 + eps))
    otol = 1e-5 if dout == "f32" else 8e-3
    assert np.abs(mean.cpu().numpy() - rmean).max() <= 1e-5 * max(1, np.abs(rmean).max())
    assert np.abs(rstd.cpu().numpy() - rrstd).max() <= 1e-5 * np.abs(rrstd).max()
    assert np.abs(y.float().cpu().numpy() - ry).max() <= otol * max(1.0, np.abs(ry).max())
    if training:   # running statistics: momentum update with the UNBIASED variance, as nn.BatchNorm
        xv = x.astype(np.float64)
        assert np.allclose(rm.cpu().numpy(), 0.9 * rm0 + 0.1 * xv.mean(0), atol=1e-5)
        assert np.allclose(rv.cpu().numpy(), 0.9 * rv0 + 0.1 * xv.var(0, ddof=1), rtol=1e-5, atol=1e-6)
    # backward against the oracle evaluated at the kernel's own arg-max (ties may differ)
    rows = P // K if K else P
    gy = rng.standard_normal((rows, C)).astype(np.float32)
    if dout == "bf16":
        gy = _bf16_round(gy)
    dx, dg, db = hip.rowbn_bwd(dev(gy).to(tdt[dout]), xd, arg, K, training, mean, rstd, dev(gamma), dev(beta),
                               slope, True)
    rdx, rdg, rdb = R.rowbn_bwd(gy, x, None if arg is None else arg.cpu().numpy(), K, training,
                                mean.cpu().numpy(), rstd.cpu().numpy(), gamma, beta, slope)
    gtol = 2e-5 if din == "f32" else 1e-2
    assert np.abs(db.cpu().numpy() - rdb).max() <= 2e-5 * max(1.0, np.abs(rdb).max())
    assert np.abs(dg.cpu().numpy() - rdg).max() <= 2e-5 * max(1.0, np.abs(rdg).max())
    assert np.abs(dx.float().cpu().numpy() - rdx).max() <= gtol * max(1.0, np.abs(rdx).max())
    if K:
        assert (arg.cpu().numpy() < K).all()
        # the same sums taken from (gy, y): arg-max pre-activations recovered from the output
        dx2, dg2, db2 = hip.rowbn_bwd(dev(gy).to(tdt[dout]), xd, arg, K, training, mean, rstd, dev(gamma),
                                      dev(beta), slope, True, y)
        ytol = 2e-5 if dout == "f32" else 1e-2
        assert np.abs(db2.cpu().numpy() - rdb).max() <= 2e-5 * max(1.0, np.abs(rdb).max())
        assert np.abs(dg2.cpu().numpy() - rdg).max() <= ytol * max(1.0, np.abs(rdg).max())
        assert np.abs(dx2.float().cpu().numpy() - rdx).max() <= max(gtol, ytol) * max(1.0, np.abs(rdx).max())


def test_rowbn_identity_statistics(hip):
    """Eval mode without statistics = activation (+max) only."""
    x = torch.randn(2 * 64 * 16, 64, device="cuda")
    y, arg = hip.rowbn_fwd(x, 16, 0.0, 0.0, False, None, None, None, None, 0.2, None, None, torch.float32)
    ref = torch.nn.functional.leaky_relu(x.view(128, 16, 64), 0.2).max(1)[0]
    assert torch.equal(y, ref)


def test_rowbn_statistics_are_bitwise_reproducible(hip):
    x = torch.randn(8 * 512 * 32, 128, device="cuda")
    outs = []
    for _ in range(3):
        mean, rstd = torch.empty(128, device="cuda"), torch.empty(128, device="cuda")
        hip.rowbn_fwd(x, 0, 1e-5, 0.1, True, None, None, None, None, 1.0, mean, rstd, torch.float32)
        outs.append((mean.clone(), rstd.clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])


# ------------------------------------------------------------------ fused spectral norm
@pytest.mark.parametrize("R_,Cn", [(64, 6), (128, 131), (256, 515), (256, 259), (1, 64), (64, 256)])
@pytest.mark.parametrize("iterate", [True, False])
def test_spectral_norm_matches_oracle(hip, R_, Cn, iterate):
    rng = np.random.default_rng(R_ + Cn)
    W = rng.standard_normal((R_, Cn)).astype(np.float32)
    u = rng.standard_normal(R_).astype(np.float32); u /= np.linalg.norm(u)
    v = rng.standard_normal(Cn).astype(np.float32); v /= np.linalg.norm(v)
    ud, vd = dev(u.copy()), dev(v.copy())
    Wsn, sigma = hip.spectral_norm_fwd(dev(W), ud, vd, iterate, 1e-12)
    rW, ru, rv, rs = R.spectral_norm_fwd(W, u, v, iterate)
    assert abs(float(sigma) - float(rs)) <= 1e-5 * abs(float(rs))
    assert np.abs(Wsn.cpu().numpy() - rW).max() <= 1e-5 * np.abs(rW).max()
    assert np.abs(ud.cpu().numpy() - ru).max() <= 1e-5 and np.abs(vd.cpu().numpy() - rv).max() <= 1e-5
    G = rng.standard_normal((R_, Cn)).astype(np.float32)
    dW = hip.spectral_norm_bwd(dev(G), Wsn, ud, vd, sigma)
    rdW = R.spectral_norm_bwd(G, rW, ru, rv, rs)
    assert np.abs(dW.cpu().numpy() - rdW).max() <= 2e-5 * max(1.0, np.abs(rdW).max())


def test_row_act_max_matches_torch(hip):
    import tpgan_amd.ops as ops
    x = torch.randn(4 * 64 * 20, 128, device="cuda", requires_grad=True)
    y = ops.row_act_max(x, 0.2, 20)
    ref = torch.nn.functional.leaky_relu(x.detach().view(256, 20, 128), 0.2).max(1)[0]
    assert torch.equal(y, ref)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().clone().requires_grad_(True)
    torch.nn.functional.leaky_relu(xr.view(256, 20, 128), 0.2).max(1)[0].backward(g)
    assert torch.allclose(x.grad, xr.grad, atol=1e-6)


def test_spectral_norm_multi_matches_chained_single_calls(hip):
    """Batched kernel (uses = 3, 2, 1) == the single-weight kernel called that many times."""
    rng = np.random.default_rng(9)
    shapes, uses = [(64, 6), (256, 515), (128, 131)], [3, 2, 1]
    Ws = [dev(rng.standard_normal(s).astype(np.float32)) for s in shapes]
    us = [torch.nn.functional.normalize(torch.randn(s[0], device="cuda"), dim=0) for s in shapes]
    vs = [torch.nn.functional.normalize(torch.randn(s[1], device="cuda"), dim=0) for s in shapes]
    u1, v1 = [u.clone() for u in us], [v.clone() for v in vs]
    flat, plan = hip.spectral_norm_multi_fwd(Ws, us, vs, uses, True, 1e-12)
    grads, singles = [], []
    for W, u, v, n, (off, st) in zip(Ws, u1, v1, uses, plan["layout"]):
        R_, Cn = W.shape
        for t in range(n):
            Wsn, sigma = hip.spectral_norm_fwd(W, u, v, True, 1e-12)
            got = flat[off + t * st: off + t * st + R_ * Cn].view(R_, Cn)
            assert torch.allclose(got, Wsn, rtol=1e-5, atol=1e-6)
            assert abs(float(flat[off + t * st + R_ * Cn + R_ + Cn]) - float(sigma)) <= 1e-5 * float(sigma)
            g = torch.randn_like(Wsn)
            grads.append(g)
            singles.append((g, Wsn, u.clone(), v.clone(), sigma))
    for a, b in zip(us + vs, u1 + v1):
        assert torch.allclose(a, b, atol=1e-6)
    dw = hip.spectral_norm_multi_bwd(flat, plan, torch.cat([g.reshape(-1) for g in grads]))
    i = 0
    for (R_, Cn), n, off in zip(shapes, uses, plan["dwoffs"]):
        ref = torch.zeros(R_, Cn, device="cuda")
        for t in range(n):
            g, Wsn, u, v, sigma = singles[i]
            ref += hip.spectral_norm_bwd(g, Wsn, u, v, sigma)
            i += 1
        assert torch.allclose(dw[off:off + R_ * Cn].view(R_, Cn), ref, rtol=1e-4, atol=1e-5)


def test_spectral_norm_split_equals_the_one_workgroup_kernel(hip):
    """The split form (rows of a weight over several workgroups, granule exchange, csrc/spectral.hip) against the
    one-workgroup kernel on the discriminators' weight shapes with up to 16 chained uses: every use's W / sigma, u, v,
    sigma and the final buffers to summation order; a second launch from the same state is bit-identical; no spin timed
    out."""
    from tpgan_amd import ops
    rng = np.random.default_rng(11)
    shapes = [(1, 64), (64, 6), (64, 256), (128, 131), (256, 259), (256, 515), (256, 515), (512, 256), (256, 512), (40, 70)]
    uses = [1, 3, 6, 2, 6, 16, 5, 6, 1, 4]
    Ws = [dev(rng.standard_normal(s).astype(np.float32) * 0.1) for s in shapes]
    u0 = [torch.nn.functional.normalize(torch.randn(s[0], device="cuda"), dim=0) for s in shapes]
    v0 = [torch.nn.functional.normalize(torch.randn(s[1], device="cuda"), dim=0) for s in shapes]
    runs = {}
    for name, split in (("split", True), ("split2", True), ("one", False)):
        ops.SN_SPLIT[0] = split
        hip._sn_plans.clear()
        us, vs = [u.clone() for u in u0], [v.clone() for v in v0]
        flat, plan = hip.spectral_norm_multi_fwd(Ws, us, vs, uses, True, 1e-12)
        torch.cuda.synchronize()
        if split:
            assert plan["split"] is not None and plan["split"]["parts"] > len(shapes)
            assert int(plan["split"]["xws"][-1]) == 0, "a bounded spin of the exchange timed out"
        runs[name] = (flat.clone(), us, vs, plan)
    ops.SN_SPLIT[0] = True
    hip._sn_plans.clear()
    a, b = runs["split"], runs["one"]
    for x, y in zip(a[1] + a[2], runs["split2"][1] + runs["split2"][2]):
        assert torch.equal(x, y)
    for (R_, Cn), n, (off, st) in zip(shapes, uses, a[3]["layout"]):
        for t in range(n):
            ga, gb = a[0][off + t * st: off + (t + 1) * st], b[0][off + t * st: off + (t + 1) * st]
            used = R_ * Cn + R_ + Cn + 1                      # (the record is padded to whole 16-byte units)
            assert torch.equal(ga[:used], runs["split2"][0][off + t * st: off + t * st + used]), (R_, Cn, t)
            assert torch.allclose(ga[:R_ * Cn + R_ + Cn], gb[:R_ * Cn + R_ + Cn], rtol=2e-5, atol=2e-6), (R_, Cn, t)
            sa, sb = float(ga[R_ * Cn + R_ + Cn]), float(gb[R_ * Cn + R_ + Cn])
            assert abs(sa - sb) <= 1e-5 * abs(sb), (R_, Cn, t)
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.allclose(x, y, atol=2e-6)


# ------------------------------------------------------------------ head: BatchNorm1d + LeakyReLU + dropout mask
@pytest.mark.parametrize("B,C,masked", [(8, 256, True), (8, 64, False), (4, 256, True), (16, 300, True), (2, 64, False)])
def test_head_bn_act_matches_oracle_and_torch_modules(hip, B, C, masked):
    """tpg_head_bn_act_fwd / _bwd against the oracle's float64 restatement (1e-5) and against the modules they replace
    -- nn.BatchNorm1d (training) -> nn.LeakyReLU -> product with a dropout mask -- outputs, running statistics, batch
    counter and all three gradients."""
    from tpgan_amd import ops
    rng = np.random.default_rng(B * 1000 + C)
    h = (rng.standard_normal((B, C)) * 2 + 0.5).astype(np.float32)
    mask = ((rng.random((B, C)) > 0.3) / 0.7).astype(np.float32) if masked else None
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(dev(rng.standard_normal(C).astype(np.float32)))
        bn.bias.copy_(dev(rng.standard_normal(C).astype(np.float32)))
        bn.running_mean.copy_(dev(rng.standard_normal(C).astype(np.float32)))
        bn.running_var.copy_(dev(rng.random(C).astype(np.float32) + 0.5))
    ref_bn = copy.deepcopy(bn)
    rm0, rv0 = bn.running_mean.cpu().numpy().copy(), bn.running_var.cpu().numpy().copy()
    x = dev(h).requires_grad_(True)
    y = ops.head_bn_act(x, bn, 0.01, None if mask is None else dev(mask))
    gy = dev(rng.standard_normal((B, C)).astype(np.float32))
    gx, gg, gb = torch.autograd.grad(y, [x, bn.weight, bn.bias], gy)
    # oracle
    ry, rmean, rrstd, rrm, rrv = R.head_bn_act_fwd(h, ref_bn.weight.detach().cpu().numpy(), ref_bn.bias.detach().cpu().numpy(),
                                                   rm0, rv0, 0.1, 1e-5, 0.01, mask)
    assert np.abs(y.detach().cpu().numpy() - ry).max() <= 1e-5 * max(1.0, np.abs(ry).max())
    assert np.abs(bn.running_mean.cpu().numpy() - rrm).max() <= 1e-6 and np.abs(bn.running_var.cpu().numpy() - rrv).max() <= 1e-5
    assert int(bn.num_batches_tracked) == 1
    rdh, rdg, rdb = R.head_bn_act_bwd(gy.cpu().numpy(), h, rmean, rrstd, ref_bn.weight.detach().cpu().numpy(),
                                      ref_bn.bias.detach().cpu().numpy(), 0.01, mask)
    for a, b in ((gx, rdh), (gg, rdg), (gb, rdb)):
        assert np.abs(a.cpu().numpy() - b).max() <= 2e-5 * max(1.0, np.abs(b).max())
    # the modules
    x2 = dev(h).requires_grad_(True)
    y2 = torch.nn.functional.leaky_relu(ref_bn(x2), 0.01)
    if mask is not None:
        y2 = y2 * dev(mask)
    g2 = torch.autograd.grad(y2, [x2, ref_bn.weight, ref_bn.bias], gy)
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
    for a, b in zip((gx, gg, gb), g2):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max() + 1))
    assert torch.allclose(bn.running_mean, ref_bn.running_mean, atol=1e-6) and torch.allclose(bn.running_var, ref_bn.running_var, atol=1e-5)
    assert int(ref_bn.num_batches_tracked) == 1


# ------------------------------------------------------------------ fused cubic interpolation
@pytest.mark.parametrize("B,Nq,Np,F,cutoff", [(2, 500, 700, 3, 0.12), (3, 1024, 4096, 3, 0.16), (1, 64, 40, 5, 0.5)])
def test_cubic_interp_matches_oracle(hip, B, Nq, Np, F, cutoff):
    rng = np.random.default_rng(Nq + Np)
    pos = rng.uniform(-0.3, 0.3, (B, Np, 3)).astype(np.float32)
    field = rng.standard_normal((B, Np, F)).astype(np.float32)
    query = rng.uniform(-0.32, 0.32, (B, Nq, 3)).astype(np.float32)
    query[0, :5] = 999.0                        # dummies: no hit
    query[0, 5:30] = pos[0, :25]                # exact coincidences
    plain, pad, hits = hip.cubic_interp(dev(query), dev(pos), dev(field), cutoff)
    rp, rpad, rh = R.cubic_interp(query, pos, field, cutoff)
    assert np.array_equal(hits.cpu().numpy(), rh)                         # same neighbour sets
    assert rh.max() == 32 or Np < 32
    for got, want in ((plain, rp), (pad, rpad)):
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert np.all(plain.cpu().numpy()[0, :5] == 0.0)


def test_cubic_interpolation_selects_padding_per_cloud(hip):
    import tpgan_amd.ops as ops
    rng = np.random.default_rng(5)
    pos = rng.uniform(-0.3, 0.3, (2, 600, 3)).astype(np.float32)
    field = rng.standard_normal((2, 600, 3)).astype(np.float32)
    query = rng.uniform(-0.3, 0.3, (2, 256, 3)).astype(np.float32)
    query[1, 0] = 999.0                         # cloud 1 gets the reference's padding edges, cloud 0 does not
    out = ops.cubic_interpolation(dev(query), dev(field), dev(pos), 0.1).cpu().numpy()
    plain, pad, hits = (t.cpu().numpy() for t in hip.cubic_interp(dev(query), dev(pos), dev(field), 0.1))
    assert np.array_equal(out[0], plain[0])
    assert np.array_equal(out[1], np.where((hits[1] < 32)[:, None], pad[1], plain[1]))
    assert (hits[1] < 32).any() and not np.array_equal(pad[1], plain[1])


@pytest.mark.parametrize("N,m", [(4096, 512), (9216, 1152), (300, 40), (20000, 64)])
def test_dataset_fps_matches_oracle(hip, N, m):
    """tpg_fps_start_f32: random first pick, every point eligible (sampling.py:50-106)."""
    rng = np.random.default_rng(N)
    pts = (rng.standard_normal((3, N, 3)) * 0.3).astype(np.float32)
    pts[0, 7] = 0.0
    start = rng.integers(0, N, 3).astype(np.int32)
    got = hip.fps(dev(pts), m, dev(start), False).cpu().numpy()
    assert np.array_equal(got, R.fps_start(pts, m, start, skip_origin=False))
    assert np.array_equal(got[:, 0], start)
    # the pointnet2 rule is unchanged
    assert np.array_equal(hip.fps(dev(pts), m).cpu().numpy(), R.fps(pts, m))


def test_dataset_fps_matches_the_reference_fixture(hip):
    """tpg_fps_start_f32 against indices produced by the reference's own sampling.py:50-106
    (tests/golden/sampling_fps.npz; capture_goldens.py `sampling`)."""
    import os
    import tpgan_amd.ops as ops
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling_fps.npz"))
    for tag in ("fluid", "dup", "origin"):
        pts, k, start = g[f"{tag}/pts"], int(g[f"{tag}/k"]), int(g[f"{tag}/start"])
        got = ops.farthest_point_sampling(dev(pts), k, initial_idx=start).cpu().numpy()
        assert np.array_equal(got, g[f"{tag}/idx"]), tag


# ------------------------------------------------------------------ grid radius search (csrc/frnn_grid.hip)
def _grid(hip, p1, p2, K, r, len1=None, len2=None):
    import tpgan_amd.ops as ops
    old = hip.GRID_MIN_POINTS, hip.GRID_MIN_PAIRS
    hip.GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = 1, 0.0             # force the grid whatever the size
    try:
        l1 = None if len1 is None else dev(np.asarray(len1, np.int64))
        l2 = None if len2 is None else dev(np.asarray(len2, np.int64))
        return hip.knn(dev(p1), dev(p2), l1, l2, K, ops.radius_sq(r), r=r)
    finally:
        hip.GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = old


@pytest.mark.parametrize("B,P1,P2,K,r", [
    (2, 512, 4096, 1, 0.0475), (2, 4096, 4096, 16, 0.035), (1, 16384, 16384, 16, 0.035), (1, 4096, 16384, 1, 0.0475),
    (3, 256, 256, 32, 2.0), (1, 100, 300, 8, 0.01), (1, 64, 64, 64, 0.2), (1, 2000, 65536, 32, 0.04), (2, 33, 77, 5, 0.3)])
def test_frnn_grid_bit_exact(hip, B, P1, P2, K, r):
    """The uniform-grid radius search against the oracle: same indices, same distances, same -1 padding."""
    rng = np.random.default_rng(P1 + P2 + K)
    p2 = fluid(rng, B, P2)
    p1 = p2.copy() if P1 == P2 else fluid(rng, B, P1, scale=0.3 * max(1.0, (P2 / 4096.0) ** (1.0 / 3.0)))   # some queries outside the box
    if P2 > 20:
        p2[0, 5] = p2[0, 7]                    # exact duplicates: ties resolved by index
        if P1 == P2:
            p1 = p2.copy()
    d, i = _grid(hip, p1, p2, K, r)
    rd, ri = R.knn(p1, p2, K, r=r)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)


def _knn_grid(hip, p1, p2, K, len1=None, len2=None):
    old = hip.KNN_GRID_MIN_POINTS, hip.GRID_MIN_PAIRS
    hip.KNN_GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = 1, 0.0         # force the grid whatever the size
    try:
        l1 = None if len1 is None else dev(np.asarray(len1, np.int64))
        l2 = None if len2 is None else dev(np.asarray(len2, np.int64))
        return hip.knn(dev(p1), dev(p2), l1, l2, K, None)
    finally:
        hip.KNN_GRID_MIN_POINTS, hip.GRID_MIN_PAIRS = old


@pytest.mark.parametrize("B,P1,P2,K", [
    (2, 4096, 4096, 20), (1, 16384, 16384, 20), (2, 4096, 4096, 1), (1, 700, 16384, 1), (1, 2000, 65536, 32),
    (3, 256, 256, 32), (1, 100, 300, 8), (1, 64, 64, 64), (2, 33, 77, 5), (1, 50, 20, 30)])
def test_knn_grid_bit_exact(hip, B, P1, P2, K):
    """Plain 3-D kNN on the uniform grid (growing cell blocks) against the oracle: indices, distances, (0, 0) padding
    when the cloud has fewer than K points; duplicates, queries outside the box."""
    rng = np.random.default_rng(P1 + 3 * P2 + K)
    p2 = fluid(rng, B, P2)
    p1 = p2.copy() if P1 == P2 else fluid(rng, B, P1, scale=0.3 * max(1.0, (P2 / 4096.0) ** (1.0 / 3.0)))
    if P2 > 20:
        p2[0, 5] = p2[0, 7]
        if P1 == P2:
            p1 = p2.copy()
    d, i = _knn_grid(hip, p1, p2, K)
    rd, ri = R.knn(p1, p2, K)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(d.cpu().numpy(), rd)


def test_knn_grid_ragged_dummies_flat_and_far(hip):
    """lengths, 999-dummies (the box grows and with it the cells: blocks must grow too), queries far outside the
    box, a cloud of identical points, a flat cloud (z = 0), and the size switch of the public entries incl. Chamfer."""
    import tpgan_amd.ops as ops
    rng = np.random.default_rng(5)
    p2 = fluid(rng, 4, 3000)
    p1 = fluid(rng, 4, 700)
    p2[1, 2500:] = 999.0
    p1[0, :5] = 50.0
    p2[2, :] = p2[2, 0]
    p2[3, :, 2] = 0.0
    l1, l2 = [700, 650, 10, 700], [3000, 2800, 3000, 7]
    d, i = _knn_grid(hip, p1, p2, 16, len1=l1, len2=l2)
    rd, ri = R.knn(p1, p2, 16, lengths1=l1, lengths2=l2)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
    big = fluid(rng, 2, 8192)
    assert 2 * 8192 * 8192 >= hip.GRID_MIN_PAIRS and 8192 >= hip.KNN_GRID_MIN_POINTS and 8192 >= hip.CHAMFER_GRID_MIN_POINTS
    dd, ii = ops.neighbour_search(dev(big), dev(big), 20)
    rd, ri = R.knn(big, big, 20)
    assert np.array_equal(ii.cpu().numpy(), ri) and np.array_equal(dd.cpu().numpy(), rd)
    other = big + rng.normal(0, 0.01, big.shape).astype(np.float32)
    d1, i1, d2, i2 = hip.chamfer_fwd(dev(big), dev(other))
    r1, j1 = R.knn(big, other, 1)
    r2, j2 = R.knn(other, big, 1)
    assert np.array_equal(i1.cpu().numpy(), j1[..., 0]) and np.array_equal(d1.cpu().numpy(), r1[..., 0])
    assert np.array_equal(i2.cpu().numpy(), j2[..., 0]) and np.array_equal(d2.cpu().numpy(), r2[..., 0])


def test_frnn_grid_ragged_dummies_and_far_queries(hip):
    """lengths1 / lengths2, 999-dummies in the searched cloud (the box grows, the cells with it: still
    exact), queries far outside the box (no neighbour: all -1), a degenerate cloud of identical points."""
    import tpgan_amd.ops as ops
    rng = np.random.default_rng(3)
    p2 = fluid(rng, 3, 3000)
    p1 = fluid(rng, 3, 700)
    p2[1, 2500:] = 999.0
    p1[0, :5] = 50.0
    p2[2, :] = p2[2, 0]
    d, i = _grid(hip, p1, p2, 16, 0.06, len1=[700, 650, 10], len2=[3000, 2800, 3000])
    rd, ri = R.knn(p1, p2, 16, lengths1=[700, 650, 10], lengths2=[3000, 2800, 3000], r=0.06)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
    assert (ri[0, :5] == -1).all()
    # through the public entry: the size switch picks the grid for a large searched cloud
    big = fluid(rng, 1, 20000)
    assert 1 * 4000 * 20000 >= hip.GRID_MIN_PAIRS
    dd, ii = ops.neighbour_search(dev(big[:, :4000]), dev(big), 8, r=0.05)
    rd, ri = R.knn(big[:, :4000], big, 8, r=0.05)
    assert np.array_equal(ii.cpu().numpy(), ri) and np.array_equal(dd.cpu().numpy(), rd)
