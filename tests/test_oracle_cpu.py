"""Oracle (oracle/tpgref.c) against the independent brute-force statement and hand-built
known-answer cases.  The reference ships no tests or golden vectors for these ops
(SURVEY.md section 4), so op-level parity is UNPINNED; these cases pin the canonical rules."""
import numpy as np
import pytest

from oracle import bruteforce as BF
from oracle import ref_ops as R


def _cloud(rng, B, N, D=3, scale=0.25):
    return rng.uniform(-scale, scale, (B, N, D)).astype(np.float32)


@pytest.mark.parametrize("D,K", [(3, 20), (32, 9), (32, 20), (64, 12), (64, 4), (5, 7)])
def test_knn_matches_bruteforce(D, K):
    rng = np.random.default_rng(D * 100 + K)
    p = rng.standard_normal((2, 150, D)).astype(np.float32)
    p[0, 5] = p[0, 7]          # exact duplicates (MSR-style repeats)
    p[1, 10:14] = p[1, 3]
    d, i = R.knn(p, p, K)
    d2, i2 = BF.knn(p, p, K)
    assert np.array_equal(i, i2) and np.array_equal(d, d2)
    assert np.array_equal(i[:, :, 0][0][:5], np.arange(5))  # self is nn #0


def test_knn_duplicates_order_by_index():
    p2 = np.zeros((1, 6, 3), np.float32)
    p2[0, :, 0] = [1, 0, 1, 0, 1, 0]
    q = np.zeros((1, 1, 3), np.float32)
    d, i = R.knn(q, p2, 6)
    assert i[0, 0].tolist() == [1, 3, 5, 0, 2, 4]
    assert d[0, 0].tolist() == [0, 0, 0, 1, 1, 1]


def test_knn_999_dummies():
    rng = np.random.default_rng(3)
    p = _cloud(rng, 1, 64)
    p[0, 40:] = 999.0  # hard-masking pads (upsampling_network.py:149)
    d, i = R.knn(p, p, 8)
    d2, i2 = BF.knn(p, p, 8)
    assert np.array_equal(i, i2)
    assert i[0, 45].tolist() == list(range(40, 48))  # ties among identical dummies -> index order


def test_knn_ragged_and_short():
    rng = np.random.default_rng(4)
    p1, p2 = _cloud(rng, 3, 20), _cloud(rng, 3, 30)
    l1, l2 = np.array([20, 7, 0]), np.array([30, 4, 9])
    d, i = R.knn(p1, p2, 6, l1, l2)
    d2, i2 = BF.knn(p1, p2, 6, l1, l2)
    assert np.array_equal(i, i2) and np.array_equal(d, d2)
    assert (i[1, :7, 4:] == 0).all() and (d[1, :7, 4:] == 0).all()  # K > len2 -> zero pad
    assert (i[2] == 0).all()                                         # len1 == 0


@pytest.mark.parametrize("K,r", [(1, 0.0475), (16, 0.035), (32, 2.0), (8, 0.01)])
def test_frnn_matches_bruteforce(K, r):
    rng = np.random.default_rng(K)
    p1, p2 = _cloud(rng, 2, 64), _cloud(rng, 2, 300)
    d, i = R.knn(p1, p2, K, r=r)
    d2, i2 = BF.knn(p1, p2, K, r=r)
    assert np.array_equal(i, i2) and np.array_equal(d, d2)
    assert ((i == -1) == (d == -1)).all()


def test_frnn_strict_radius():
    q = np.zeros((1, 1, 3), np.float32)
    p2 = np.array([[[0.5, 0, 0], [0.25, 0, 0], [0.5000001, 0, 0]]], np.float32)
    d, i = R.knn(q, p2, 3, r=0.5)  # d^2 == r^2 is OUT (strict <)
    assert i[0, 0].tolist() == [1, -1, -1]
    assert d[0, 0].tolist() == [0.0625, -1, -1]


@pytest.mark.parametrize("N,m", [(500, 64), (64, 64), (10, 25)])
def test_fps_matches_bruteforce(N, m):
    rng = np.random.default_rng(N)
    x = _cloud(rng, 2, N)
    x[0, 3] = 0.0          # origin point: never eligible (|x|^2 <= 1e-3)
    x[1, :3] = 999.0       # dummies
    assert np.array_equal(R.fps(x, m), BF.fps(x, m))


def test_fps_known_answers():
    # 4 collinear points; start at 0; farthest from 0 is 3; then 1 vs 2: temp = min dist to {0,3}
    x = np.array([[[1, 0, 0], [2, 0, 0], [3.5, 0, 0], [5, 0, 0]]], np.float32)
    assert R.fps(x, 4)[0].tolist() == [0, 3, 2, 1]
    # duplicates: ties -> smallest index
    x = np.array([[[1, 0, 0], [3, 0, 0], [3, 0, 0], [3, 0, 0]]], np.float32)
    assert R.fps(x, 3)[0].tolist() == [0, 1, 0]  # tie 1/2/3 -> 1; then every temp is 0 -> smallest index
    # all points within the origin ball: nothing eligible -> index 0 repeated
    x = np.full((1, 5, 3), 0.01, np.float32)
    assert R.fps(x, 4)[0].tolist() == [0, 0, 0, 0]


def test_fps_more_samples_than_points_does_not_crash():
    rng = np.random.default_rng(0)
    x = _cloud(rng, 1, 8) + 1.0
    out = R.fps(x, 20)
    assert out.shape == (1, 20) and (out >= 0).all() and (out < 8).all()


@pytest.mark.parametrize("r,ns", [(0.1, 16), (0.15, 32), (0.6, 16), (0.005, 8)])
def test_ball_query_matches_bruteforce(r, ns):
    rng = np.random.default_rng(ns)
    x = _cloud(rng, 2, 400)
    q = x[:, ::5].copy()
    assert np.array_equal(R.ball_query(r, ns, x, q), BF.ball_query(r, ns, x, q))


def test_ball_query_known_answers():
    x = np.zeros((1, 6, 3), np.float32)
    x[0, :, 0] = [5, 0.1, 5, 0.2, 0.3, 5]
    q = np.zeros((1, 2, 3), np.float32)
    q[0, 1, 0] = 100.0
    idx = R.ball_query(1.0, 4, x, q)
    assert idx[0, 0].tolist() == [1, 3, 4, 1]   # index order; tail filled with FIRST hit
    assert idx[0, 1].tolist() == [0, 0, 0, 0]   # no hit -> zeros
    idx = R.ball_query(1.0, 2, x, q)
    assert idx[0, 0].tolist() == [1, 3]         # stops at nsample


def test_group_gather_roundtrip():
    rng = np.random.default_rng(1)
    f = rng.standard_normal((2, 7, 100)).astype(np.float32)
    idx = rng.integers(0, 100, (2, 30, 5)).astype(np.int32)
    idx[0, 0] = 7  # repeated index
    out = R.group_fwd(f, idx)
    assert np.array_equal(out, BF.group_fwd(f, idx))
    g = rng.standard_normal(out.shape).astype(np.float32)
    assert np.allclose(R.group_bwd(g, idx, 100), BF.group_bwd(g, idx, 100), atol=1e-5)
    gi = idx[:, :, 0].copy()
    assert np.array_equal(R.gather_fwd(f, gi), BF.gather_fwd(f, gi))
    gg = rng.standard_normal((2, 7, 30)).astype(np.float32)
    ref = np.zeros((2, 7, 100))
    for b in range(2):
        for c in range(7):
            np.add.at(ref[b, c], gi[b], gg[b, c])
    assert np.allclose(R.gather_bwd(gg, gi, 100), ref, atol=1e-5)


def test_chamfer_hand_computed():
    s = np.array([[[0, 0, 0], [1, 0, 0]]], np.float32)
    t = np.array([[[0, 0, 0.5], [1, 0, 0], [4, 0, 0]]], np.float32)
    d1, i1, d2, i2 = R.chamfer_fwd(s, t)
    assert d1[0].tolist() == [0.25, 0.0] and i1[0].tolist() == [0, 1]
    assert d2[0].tolist() == [0.25, 0.0, 9.0] and i2[0].tolist() == [0, 1, 1]
    gs, gt = R.chamfer_bwd(s, t, i1, i2, np.ones_like(d1), np.ones_like(d2))
    # d(sum)/ds0 = 2(s0-t0) [fwd] + 2(s0-t0) [bwd term of t0]
    assert np.allclose(gs[0, 0], [0, 0, -2.0])
    assert np.allclose(gs[0, 1], [-6.0, 0, 0])       # t2 pulls s1: -2(t2 - s1)
    assert np.allclose(gt[0, 2], [6.0, 0, 0])


def test_chamfer_matches_bruteforce_value_and_numeric_grad():
    rng = np.random.default_rng(5)
    s, t = _cloud(rng, 2, 40), _cloud(rng, 2, 55)
    d1, i1, d2, i2 = R.chamfer_fwd(s, t)
    val = d1.sum(1).mean() + d2.sum(1).mean()
    assert abs(val - BF.chamfer(s, t)) < 1e-5
    gs, _ = R.chamfer_bwd(s, t, i1, i2, np.full_like(d1, 0.5), np.full_like(d2, 0.5))
    eps = 1e-3
    s2 = s.copy(); s2[1, 3, 1] += eps
    num = (BF.chamfer(s2, t) - BF.chamfer(s, t)) / eps
    assert abs(num - gs[1, 3, 1]) < 5e-3


def test_three_nn_and_interpolate():
    rng = np.random.default_rng(6)
    u, k = _cloud(rng, 2, 30), _cloud(rng, 2, 50)
    d2, idx = R.three_nn(u, k)
    bd, bi = BF.three_nn(u, k)
    assert np.array_equal(idx, bi) and np.array_equal(d2, bd)
    f = rng.standard_normal((2, 4, 50)).astype(np.float32)
    w = rng.uniform(0, 1, (2, 30, 3)).astype(np.float32)
    out = R.three_interp_fwd(f, idx, w)
    ref = np.stack([(f[b][:, idx[b]] * w[b][None]).sum(-1) for b in range(2)])
    assert np.allclose(out, ref, atol=1e-5)
    g = rng.standard_normal(out.shape).astype(np.float32)
    gf = R.three_interp_bwd(g, idx, w, 50)
    ref = np.zeros((2, 4, 50))
    for b in range(2):
        for c in range(4):
            np.add.at(ref[b, c], idx[b].reshape(-1), (g[b, c][:, None] * w[b]).reshape(-1))
    assert np.allclose(gf, ref, atol=1e-5)


# ------------------------------------------------------------------ cubic interpolation (--use_vel)
@pytest.mark.parametrize("case", ["dense", "with_far_queries", "coincident"])
def test_cubic_interpolation_matches_the_reference_algorithm(case, oracle_cpu):
    """ops.cubic_interpolation (one fused search-and-sum per batch + the padding selection) against
    the reference's algorithm written out as an edge list (oracle/bruteforce.py): FRNN-32, unique,
    FRNN again, kNN-4 padding multigraph, DGL-style scatter sums.  PARITY UNPINNED against DGL itself
    (not installable here); the edge-list statement follows gcn_lib/interpolation.py line by line."""
    import torch
    import tpgan_amd.ops as ops
    rng = np.random.default_rng({"dense": 0, "with_far_queries": 1, "coincident": 2}[case])
    B, Np, Nq, F, cutoff = 2, 300, 96, 3, 0.16
    pos = rng.uniform(-0.3, 0.3, (B, Np, 3)).astype(np.float32)
    field = rng.standard_normal((B, Np, F)).astype(np.float32)
    query = rng.uniform(-0.3, 0.3, (B, Nq, 3)).astype(np.float32)
    if case == "with_far_queries":            # 999-dummies of a masked prediction: no hit -> padding active
        query[0, :7] = 999.0
        query[0, 7:20] = rng.uniform(0.36, 0.4, (13, 3))          # few hits
    if case == "coincident":
        query[:, :40] = pos[:, :40]                                # d = 0 exactly
    got = ops.cubic_interpolation(torch.from_numpy(query), torch.from_numpy(field), torch.from_numpy(pos), cutoff).numpy()
    for b in range(B):
        want = BF.cubic_interpolation(query[b], field[b], pos[b], cutoff)
        assert np.abs(got[b] - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), (case, b)
    if case == "with_far_queries":
        assert np.all(got[0, :7] == 0.0)                           # no neighbours: 0 / (0 + 1e-6)
    # the per-sample 2-D signature of the reference
    one = ops.cubic_interpolation(torch.from_numpy(query[1]), torch.from_numpy(field[1]), torch.from_numpy(pos[1]), cutoff)
    assert np.array_equal(one.numpy(), got[1])


def test_dataset_fps_random_start_no_origin_skip(oracle_cpu):
    """sampling.py:50-106 restated in numpy (squared fp32 distances, argmax = first maximum, start
    index given, NO |x|^2 > 1e-3 rule) against the oracle's start variant and ops.farthest_point_sampling."""
    import torch
    import tpgan_amd.ops as ops
    rng = np.random.default_rng(4)
    pts = rng.standard_normal((2, 700, 3)).astype(np.float32) * 0.2
    pts[0, 5] = 0.0                                    # at the origin: eligible here, skipped by pointnet2's rule
    pts[1, :40] *= 0.05
    start = np.array([17, 300], np.int32)
    got = R.fps_start(pts, 64, start, skip_origin=False)
    for b in range(2):
        x = pts[b]
        idx = [int(start[b])]
        mind = ((x - x[idx[0]]) ** 2).sum(-1, dtype=np.float32)
        for _ in range(63):
            j = int(np.argmax(mind))
            idx.append(j)
            mind = np.minimum(mind, ((x - x[j]) ** 2).sum(-1, dtype=np.float32))
        assert np.array_equal(got[b], np.array(idx)), b
    t = ops.farthest_point_sampling(torch.from_numpy(pts), 64, initial_idx=start.tolist())
    assert np.array_equal(t.numpy(), got)
    one = ops.farthest_point_sampling(torch.from_numpy(pts[1]), 64, initial_idx=300)
    assert np.array_equal(one.numpy(), got[1])
    d = ops.sample_patch_with_fps(torch.from_numpy(pts[0]), 256, seed_idx=3, initial_idx=0)
    assert d["patch_pos"].shape == (256, 3) and d["ds_pos"].shape == (32, 3) and int(d["patch_idx"][0]) == 3


def test_dataset_fps_oracle_matches_the_reference_fixture():
    """f4 pinned to the reference: tests/golden/sampling_fps.npz holds the indices
    `/root/reference/sampling.py:50-106` itself returned (capture_goldens.py, numba shim) for
    three clouds incl. exact duplicates, a point at the origin and a run of repeats."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampling_fps.npz"))
    for tag in ("fluid", "dup", "origin"):
        pts, k, start = g[f"{tag}/pts"], int(g[f"{tag}/k"]), int(g[f"{tag}/start"])
        got = R.fps_start(pts[None], k, np.array([start], np.int32), skip_origin=False)[0]
        assert np.array_equal(got, g[f"{tag}/idx"]), tag
