"""Independent numpy statement of the op DEFINITIONS.  TEST INFRASTRUCTURE ONLY.

Written from the definitions (full distance matrix + lexsort, python loops for
the sequential ops), not from tpgref.c, so the two can cross-check each other
(oracle construction step 1).  Small sizes only.
"""
import numpy as np


def sqdist_matrix(a, b):
    """(P1,D),(P2,D) -> (P1,P2) fp32, sum over d in order, mul and add rounded
    separately (numpy ufuncs never fuse)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    acc = np.zeros((a.shape[0], b.shape[0]), np.float32)
    for d in range(a.shape[1]):
        t = a[:, d][:, None] - b[:, d][None, :]
        acc = acc + t * t
    return acc


def knn(p1, p2, K, lengths1=None, lengths2=None, r=None):
    p1 = np.asarray(p1, np.float32)
    p2 = np.asarray(p2, np.float32)
    B, P1, _ = p1.shape
    P2 = p2.shape[1]
    radius = r is not None
    if radius:
        r32 = np.float32(r)
        r2 = np.float32(r32 * r32)
    dist = np.full((B, P1, K), -1.0 if radius else 0.0, np.float32)
    idx = np.full((B, P1, K), -1 if radius else 0, np.int64)
    for b in range(B):
        n1 = P1 if lengths1 is None else int(lengths1[b])
        n2 = P2 if lengths2 is None else int(lengths2[b])
        if n1 == 0 or n2 == 0:
            continue
        dm = sqdist_matrix(p1[b, :n1], p2[b, :n2])
        for i in range(n1):
            row = dm[i]
            order = np.lexsort((np.arange(n2), row))  # primary dist, secondary idx
            if radius:
                order = order[row[order] < r2]
            order = order[:K]
            dist[b, i, :len(order)] = row[order]
            idx[b, i, :len(order)] = order
    return dist, idx


def fps(xyz, m):
    xyz = np.asarray(xyz, np.float32)
    B, N, _ = xyz.shape
    out = np.zeros((B, m), np.int32)
    for b in range(B):
        x = xyz[b]
        mag = (x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]) + x[:, 2] * x[:, 2]
        ok = mag > np.float32(1e-3)
        temp = np.full(N, 1e10, np.float32)
        old = 0
        for j in range(1, m):
            d = sqdist_matrix(x, x[old:old + 1])[:, 0]
            temp = np.where(ok, np.minimum(temp, d), temp)
            cand = np.where(ok, temp, np.float32(-np.inf))
            best = cand.max() if ok.any() else -np.inf
            old = int(np.argmax(cand)) if best > -1.0 else 0  # first max = smallest idx
            out[b, j] = old
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    xyz = np.asarray(xyz, np.float32)
    new_xyz = np.asarray(new_xyz, np.float32)
    B, S, _ = new_xyz.shape
    r32 = np.float32(radius)
    r2 = np.float32(r32 * r32)
    out = np.zeros((B, S, nsample), np.int32)
    for b in range(B):
        dm = sqdist_matrix(new_xyz[b], xyz[b])
        for s in range(S):
            hits = np.nonzero(dm[s] < r2)[0][:nsample]
            if len(hits):
                out[b, s, :] = hits[0]
                out[b, s, :len(hits)] = hits
    return out


def group_fwd(feat, idx):
    feat = np.asarray(feat, np.float32)
    B = feat.shape[0]
    return np.stack([feat[b][:, idx[b]] for b in range(B)])


def group_bwd(gout, idx, N):
    gout = np.asarray(gout, np.float64)
    B, C, S, K = gout.shape
    g = np.zeros((B, C, N), np.float64)
    for b in range(B):
        for c in range(C):
            np.add.at(g[b, c], idx[b].reshape(-1), gout[b, c].reshape(-1))
    return g


def gather_fwd(feat, idx):
    feat = np.asarray(feat, np.float32)
    return np.stack([feat[b][:, idx[b]] for b in range(feat.shape[0])])


def chamfer(src, tgt):
    """sum over points, mean over batch, both directions (float64 reference value)."""
    tot = 0.0
    for b in range(src.shape[0]):
        dm = sqdist_matrix(src[b], tgt[b]).astype(np.float64)
        tot += dm.min(1).sum() + dm.min(0).sum()
    return tot / src.shape[0]


def three_nn(unknown, known):
    d, i = knn(unknown, known, 3)
    return d, i.astype(np.int32)


# ---- cubic interpolation: the reference's algorithm step by step ---------------------------
def _frnn_lists(query, cand, r, K=32):
    """Per query: indices of the <=K nearest candidates with d^2 < fp32(r)^2, ascending (d^2, idx)."""
    r2 = np.float32(np.float32(r) * np.float32(r))
    out = []
    for q in query.astype(np.float32):
        d = ((q[None, :] - cand.astype(np.float32)) ** 2)
        d2 = (d[:, 0] + d[:, 1]) + d[:, 2]                     # fp32, d order
        order = np.lexsort((np.arange(len(cand)), d2))
        out.append([int(j) for j in order if d2[j] < r2][:K])
    return out


def _bicubic(r, cutoff):
    coeff = 8. / (np.pi * cutoff ** 3)
    q = (r / np.float32(cutoff)).astype(np.float32)
    ker = np.zeros_like(q)
    m1 = (q >= 0) & (q <= 0.5)
    m2 = (q > 0.5) & (q <= 1)
    ker[m1] = (6. * (q ** 3 - q ** 2) + 1.)[m1]
    ker[m2] = (2. * (1. - q) ** 3)[m2]
    return (ker * np.float32(coeff)).astype(np.float32)


def cubic_interpolation(query, field, pos, cutoff):
    """gcn_lib/interpolation.py:107-123 + :16-75 for ONE sample, as a literal edge list (a multigraph:
    the kNN-4 padding adds edges without removing duplicates).  query (Nq,3), field (Np,F), pos (Np,3)."""
    query, field, pos = (np.asarray(a, np.float32) for a in (query, field, pos))
    first = _frnn_lists(query, pos, cutoff)
    in_range = np.unique(np.array([j for l in first for j in l], dtype=np.int64))   # filter_out_of_range
    cand, fld = pos[in_range], field[in_range]
    lists = _frnn_lists(query, cand, cutoff)
    src = [j for l in lists for j in l]
    dst = [q for q, l in enumerate(lists) for _ in l]
    if any(len(l) == 0 for l in lists):                                                 # knn_padding
        for q, l in enumerate(lists):
            if len(l) < 32:
                d = ((query[q][None, :] - cand) ** 2)
                d2 = (d[:, 0] + d[:, 1]) + d[:, 2]
                near = np.lexsort((np.arange(len(cand)), d2))[:4]
                src += [int(j) for j in near]
                dst += [q] * len(near)
    out = np.zeros((len(query), field.shape[1]), np.float64)
    k = np.zeros(len(query), np.float64)
    if src:
        src, dst = np.array(src), np.array(dst)
        a, b = cand[src], query[dst]
        dist = (a ** 2 + b ** 2 - 2 * b * a).sum(1)                                   # l2dist :11-14
        dist[dist < 1e-8] = 0.
        w = _bicubic(np.sqrt(dist.astype(np.float32)), cutoff).astype(np.float64)
        np.add.at(out, dst, w[:, None] * fld[src].astype(np.float64))
        np.add.at(k, dst, w)
    return (out / (k[:, None] + 1e-6)).astype(np.float32)
