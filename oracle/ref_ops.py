"""numpy front-end of libtpgref.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY.

Every function takes/returns numpy arrays with the layouts of include/tpgan_ops.h.
"""
import ctypes as C

import numpy as np

from . import build

_lib = None

_F = C.POINTER(C.c_float)
_I32 = C.POINTER(C.c_int32)
_I64 = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _f(a):
    return a.ctypes.data_as(_F)


def _i32(a):
    return a.ctypes.data_as(_I32)


def _i64(a):
    return None if a is None else a.ctypes.data_as(_I64)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f"oracle {name} failed with status {rc}")


def radius_sq(r):
    """r (python float) -> fp32 r*r exactly as the product's host code does."""
    r32 = np.float32(r)
    return float(np.float32(r32 * r32))


def knn(p1, p2, K, lengths1=None, lengths2=None, r=None):
    p1, p2 = _c(p1, np.float32), _c(p2, np.float32)
    B, P1, D = p1.shape
    P2 = p2.shape[1]
    l1 = None if lengths1 is None else _c(lengths1, np.int64)
    l2 = None if lengths2 is None else _c(lengths2, np.int64)
    dist = np.empty((B, P1, K), np.float32)
    idx = np.empty((B, P1, K), np.int64)
    r2 = -1.0 if r is None else radius_sq(r)
    _chk(lib().tpgref_knn_f32(_f(p1), _f(p2), _i64(l1), _i64(l2), B, P1, P2, D, K,
                              C.c_float(r2), _f(dist), _i64(idx)), "knn")
    return dist, idx


def cubic_interp(query, pos, field, cutoff):
    """-> (out_plain (B,Nq,F), out_pad (B,Nq,F), hits (B,Nq) i32); include/tpgan_ops.h."""
    query, pos, field = _c(query, np.float32), _c(pos, np.float32), _c(field, np.float32)
    B, Nq, _ = query.shape
    Np, F = pos.shape[1], field.shape[2]
    plain, pad = np.empty((B, Nq, F), np.float32), np.empty((B, Nq, F), np.float32)
    hits = np.empty((B, Nq), np.int32)
    _chk(lib().tpgref_cubic_interp_f32(_f(query), _f(pos), _f(field), B, Nq, Np, F, C.c_float(cutoff),
                                       _f(plain), _f(pad), _i32(hits)), "cubic_interp")
    return plain, pad, hits


def chamfer_fwd(src, tgt):
    src, tgt = _c(src, np.float32), _c(tgt, np.float32)
    B, N, _ = src.shape
    M = tgt.shape[1]
    d1, i1 = np.empty((B, N), np.float32), np.empty((B, N), np.int64)
    d2, i2 = np.empty((B, M), np.float32), np.empty((B, M), np.int64)
    _chk(lib().tpgref_chamfer_fwd_f32(_f(src), _f(tgt), B, N, M, _f(d1), _i64(i1),
                                      _f(d2), _i64(i2)), "chamfer_fwd")
    return d1, i1, d2, i2


def chamfer_bwd(src, tgt, i1, i2, g1, g2):
    src, tgt = _c(src, np.float32), _c(tgt, np.float32)
    i1, i2 = _c(i1, np.int64), _c(i2, np.int64)
    g1, g2 = _c(g1, np.float32), _c(g2, np.float32)
    B, N, _ = src.shape
    M = tgt.shape[1]
    gs, gt = np.empty_like(src), np.empty_like(tgt)
    _chk(lib().tpgref_chamfer_bwd_f32(_f(src), _f(tgt), B, N, M, _i64(i1), _i64(i2),
                                      _f(g1), _f(g2), _f(gs), _f(gt)), "chamfer_bwd")
    return gs, gt


def fps(xyz, m):
    xyz = _c(xyz, np.float32)
    B, N, _ = xyz.shape
    temp = np.empty((B, N), np.float32)
    idx = np.empty((B, m), np.int32)
    _chk(lib().tpgref_fps_f32(_f(xyz), B, N, m, _f(temp), _i32(idx)), "fps")
    return idx


def fps_start(xyz, m, start=None, skip_origin=False):
    xyz = _c(xyz, np.float32)
    B, N, _ = xyz.shape
    temp = np.empty((B, N), np.float32)
    idx = np.empty((B, m), np.int32)
    st = None if start is None else _c(start, np.int32)
    _chk(lib().tpgref_fps_start_f32(_f(xyz), None if st is None else _i32(st), int(bool(skip_origin)), B, N, m,
                                    _f(temp), _i32(idx)), "fps_start")
    return idx


def gather_fwd(feat, idx):
    feat, idx = _c(feat, np.float32), _c(idx, np.int32)
    B, Cc, N = feat.shape
    S = idx.shape[1]
    out = np.empty((B, Cc, S), np.float32)
    _chk(lib().tpgref_gather_fwd_f32(_f(feat), _i32(idx), B, Cc, N, S, _f(out)), "gather_fwd")
    return out


def gather_bwd(gout, idx, N):
    gout, idx = _c(gout, np.float32), _c(idx, np.int32)
    B, Cc, S = gout.shape
    g = np.empty((B, Cc, N), np.float32)
    _chk(lib().tpgref_gather_bwd_f32(_f(gout), _i32(idx), B, Cc, N, S, _f(g)), "gather_bwd")
    return g


def gather_rows_fwd(rows, idx):
    """tpg_gather_rows_fwd_f32 = gather_fwd on the transposed tensors."""
    return np.ascontiguousarray(gather_fwd(np.ascontiguousarray(np.transpose(rows, (0, 2, 1))), idx).transpose(0, 2, 1))


def gather_rows_bwd(gout, idx, N):
    return np.ascontiguousarray(gather_bwd(np.ascontiguousarray(np.transpose(gout, (0, 2, 1))), idx, N).transpose(0, 2, 1))


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _c(xyz, np.float32), _c(new_xyz, np.float32)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = np.empty((B, S, nsample), np.int32)
    _chk(lib().tpgref_ball_query_f32(_f(xyz), _f(new_xyz), B, N, S, C.c_float(radius),
                                     nsample, _i32(idx)), "ball_query")
    return idx


def group_fwd(feat, idx):
    feat, idx = _c(feat, np.float32), _c(idx, np.int32)
    B, Cc, N = feat.shape
    _, S, K = idx.shape
    out = np.empty((B, Cc, S, K), np.float32)
    _chk(lib().tpgref_group_fwd_f32(_f(feat), _i32(idx), B, Cc, N, S, K, _f(out)), "group_fwd")
    return out


def group_bwd(gout, idx, N):
    gout, idx = _c(gout, np.float32), _c(idx, np.int32)
    B, Cc, S, K = gout.shape
    g = np.empty((B, Cc, N), np.float32)
    _chk(lib().tpgref_group_bwd_f32(_f(gout), _i32(idx), B, Cc, N, S, K, _f(g)), "group_bwd")
    return g


def three_nn(unknown, known):
    unknown, known = _c(unknown, np.float32), _c(known, np.float32)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((B, n, 3), np.float32)
    idx = np.empty((B, n, 3), np.int32)
    _chk(lib().tpgref_three_nn_f32(_f(unknown), _f(known), B, n, m, _f(d2), _i32(idx)), "three_nn")
    return d2, idx


def three_interp_fwd(feat, idx, w):
    feat, idx, w = _c(feat, np.float32), _c(idx, np.int32), _c(w, np.float32)
    B, Cc, m = feat.shape
    n = idx.shape[1]
    out = np.empty((B, Cc, n), np.float32)
    _chk(lib().tpgref_three_interp_fwd_f32(_f(feat), _i32(idx), _f(w), B, Cc, m, n, _f(out)),
         "three_interp_fwd")
    return out


def three_interp_bwd(gout, idx, w, m):
    gout, idx, w = _c(gout, np.float32), _c(idx, np.int32), _c(w, np.float32)
    B, Cc, n = gout.shape
    g = np.empty((B, Cc, m), np.float32)
    _chk(lib().tpgref_three_interp_bwd_f32(_f(gout), _i32(idx), _f(w), B, Cc, m, n, _f(g)),
         "three_interp_bwd")
    return g


def rowcombine_fwd(U, QE, idx, mode, slope=0.2):
    U, idx = _c(U, np.float32), _c(idx, np.int32)
    QE = None if QE is None else _c(QE, np.float32)
    B, N, Cc = U.shape
    _, S, K = idx.shape
    out = np.empty((B, S, K, Cc), np.float32)
    _chk(lib().tpgref_rowcombine_fwd_f32(_f(U), None if QE is None else _f(QE), _i32(idx), mode, B, N, S,
                                         K, Cc, C.c_float(slope), _f(out)), "rowcombine_fwd")
    return out


def rowcombine_bwd(gout, idx, E, mode, N, slope=0.2):
    gout, idx = _c(gout, np.float32), _c(idx, np.int32)
    E = None if E is None else _c(E, np.float32)
    B, S, K, Cc = gout.shape
    gU = np.empty((B, N, Cc), np.float32)
    gQE = np.empty((B, S, Cc), np.float32) if mode != 0 else None
    _chk(lib().tpgref_rowcombine_bwd_f32(_f(gout), _i32(idx), None if E is None else _f(E), mode, B, N, S, K,
                                         Cc, C.c_float(slope), _f(gU), None if gQE is None else _f(gQE)),
         "rowcombine_bwd")
    return gU, gQE


def rowcombine_edge_fwd(Y, idx, slope_a=0.2, slope_e=0.2):
    """The EdgeConv front end on one product (include/tpgan_ops.h, tpg_rowcombine_edge_fwd; reference
    gcn_lib/pointnet/gcn.py:176-180,207-210): Y (B,N,2C) = f [We; Wn]^T; restated over rowcombine_fwd(mode EDGE)."""
    Y = _c(Y, np.float32)
    Cc = Y.shape[2] // 2
    A = Y[:, :, Cc:]
    A = np.where(A > 0, A, A * np.float32(slope_a)).astype(np.float32)
    return rowcombine_fwd(A, Y[:, :, :Cc], idx, 2, slope_e)


def rowcombine_edge_bwd(gout, idx, Y, slope_a=0.2, slope_e=0.2):
    Y = _c(Y, np.float32)
    Cc = Y.shape[2] // 2
    gU, gE = rowcombine_bwd(gout, idx, Y[:, :, :Cc], 2, Y.shape[1], slope_e)
    gA = np.where(Y[:, :, Cc:] > 0, gU, gU * np.float32(slope_a)).astype(np.float32)
    return np.concatenate([gE, gA], axis=2)


# ---- a head's BatchNorm1d + LeakyReLU + dropout mask (include/tpgan_ops.h, tpg_head_bn_act_*): numpy restatement
# (float64 inside) of nn.BatchNorm1d in training mode -> nn.LeakyReLU -> x * mask, reference discriminator.py:503-516.
def head_bn_act_fwd(h, gamma, beta, running_mean, running_var, momentum, eps, slope, mask):
    h64 = np.asarray(h, np.float64)
    B = h64.shape[0]
    mean = h64.mean(0)
    var = h64.var(0)
    rstd = 1.0 / np.sqrt(var + eps)
    z = (h64 - mean) * rstd * (1.0 if gamma is None else np.asarray(gamma, np.float64)) + \
        (0.0 if beta is None else np.asarray(beta, np.float64))
    y = np.where(z > 0, z, z * slope)
    if mask is not None:
        y = y * np.asarray(mask, np.float64)
    new_rm = None if running_mean is None else (1 - momentum) * np.asarray(running_mean, np.float64) + momentum * mean
    new_rv = None if running_var is None else \
        (1 - momentum) * np.asarray(running_var, np.float64) + momentum * var * B / (B - 1)
    f = lambda a: None if a is None else a.astype(np.float32)
    return f(y), f(mean), f(rstd), f(new_rm), f(new_rv)


def head_bn_act_bwd(gy, h, mean, rstd, gamma, beta, slope, mask):
    h64, g64 = np.asarray(h, np.float64), np.asarray(gy, np.float64)
    B = h64.shape[0]
    ga = 1.0 if gamma is None else np.asarray(gamma, np.float64)
    be = 0.0 if beta is None else np.asarray(beta, np.float64)
    xh = (h64 - np.asarray(mean, np.float64)) * np.asarray(rstd, np.float64)
    gz = g64 if mask is None else g64 * np.asarray(mask, np.float64)
    gz = np.where(xh * ga + be > 0, gz, gz * slope)
    db, dg = gz.sum(0), (gz * xh).sum(0)
    dh = ga * np.asarray(rstd, np.float64) * (gz - db / B - xh * dg / B)
    return dh.astype(np.float32), dg.astype(np.float32), db.astype(np.float32)


# ---- fused BatchNorm + LeakyReLU (+ max over K) on rows: numpy restatement (float64 inside) of the
# build's own fused form of [BatchNorm2d -> (Leaky)ReLU -> max over nsample]
# (reference discriminator.py:63-78,145-150,279-282); equality with the reference is pinned at
# model level by tests/golden.
def _bn_preact32(x, mean, rstd, gamma, beta):
    """z = (x - mean) * (gamma * rstd) + beta in fp32, op for op as csrc/rowbn.hip::bn_z, so the
    sign of z (the LeakyReLU mask) is the one the kernel sees."""
    x32 = np.asarray(x, np.float32)
    Cc = x32.shape[1]
    g = np.ones(Cc, np.float32) if gamma is None else np.asarray(gamma, np.float32)
    b = np.zeros(Cc, np.float32) if beta is None else np.asarray(beta, np.float32)
    a = g * np.asarray(rstd, np.float32)
    return (x32 - np.asarray(mean, np.float32)) * a + b


def rowbn_fwd(x, K, eps, gamma, beta, slope, training=True, mean=None, rstd=None):
    x64 = np.asarray(x, np.float64)
    P, Cc = x64.shape
    if training:
        mean = x64.mean(0)
        rstd = 1.0 / np.sqrt(x64.var(0) + eps)
    mean, rstd = np.asarray(mean, np.float32), np.asarray(rstd, np.float32)
    z = _bn_preact32(x, mean, rstd, gamma, beta)
    y = np.where(z > 0, z, z * np.float32(slope))
    arg = None
    if K:
        yk = y.reshape(P // K, K, Cc)
        arg = yk.argmax(1).astype(np.uint8)          # first maximum
        y = yk.max(1)
    return y.astype(np.float32), mean, rstd, arg


def rowbn_bwd(gy, x, arg, K, training, mean, rstd, gamma, beta, slope):
    x64 = np.asarray(x, np.float64)
    P, Cc = x64.shape
    g_ = np.ones(Cc) if gamma is None else np.asarray(gamma, np.float64)
    z = _bn_preact32(x, mean, rstd, gamma, beta)
    mean, rstd = np.asarray(mean, np.float64), np.asarray(rstd, np.float64)
    xhat = (x64 - mean) * rstd
    gy64 = np.asarray(gy, np.float64)
    if K:
        full = np.zeros((P // K, K, Cc))
        gi, ci = np.meshgrid(np.arange(P // K), np.arange(Cc), indexing="ij")
        full[gi, arg.astype(np.int64), ci] = gy64
        gy64 = full.reshape(P, Cc)
    g = gy64 * np.where(z > 0, 1.0, slope)
    dbeta, dgamma = g.sum(0), (g * xhat).sum(0)
    if training:
        dx = g_ * rstd * (g - dbeta / P - xhat * dgamma / P)
    else:
        dx = g_ * rstd * g
    return dx.astype(np.float32), dgamma.astype(np.float32), dbeta.astype(np.float32)


# ---- spectral norm (torch.nn.utils.spectral_norm's forward pre-hook, n_power_iterations = 1) ----
def spectral_norm_fwd(W, u, v, iterate, eps=1e-12):
    W64, u64, v64 = np.asarray(W, np.float64), np.asarray(u, np.float64), np.asarray(v, np.float64)
    if iterate:
        t = W64.T @ u64
        v64 = t / max(np.linalg.norm(t), eps)
        s = W64 @ v64
        u64 = s / max(np.linalg.norm(s), eps)
    sigma = u64 @ (W64 @ v64)
    return (W64 / sigma).astype(np.float32), u64.astype(np.float32), v64.astype(np.float32), np.float32(sigma)


def spectral_norm_bwd(G, Wsn, u, v, sigma):
    G64, W64 = np.asarray(G, np.float64), np.asarray(Wsn, np.float64)
    d = (G64 * W64).sum()
    return ((G64 - d * np.outer(u, v)) / float(sigma)).astype(np.float32)
