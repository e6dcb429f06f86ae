"""CPU-tensor op backend built on the C restatement.  TEST INFRASTRUCTURE ONLY.

`install()` registers it with `tpgan_amd.ops` for CPU tensors so that (a) tests can
run the host-side model code on CPU, (b) the golden-capture script can run the
reference's Python over it in the build container, and (c) bench.py's
`cpu_baseline` leg can time the same step function on the host cores.  The
product never calls `install()`.
"""
import numpy as np
import torch

from . import ref_ops as R


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class OracleBackend:
    name = "oracle-cpu"

    def knn(self, p1, p2, len1, len2, K, r2):
        # ref_ops.knn squares r itself; pass sqrt-free by calling the C symbol directly
        import ctypes as C
        a, b = R._c(_np(p1), np.float32), R._c(_np(p2), np.float32)
        B, P1, D = a.shape
        P2 = b.shape[1]
        l1 = None if len1 is None else R._c(_np(len1), np.int64)
        l2 = None if len2 is None else R._c(_np(len2), np.int64)
        dist = np.empty((B, P1, K), np.float32)
        idx = np.empty((B, P1, K), np.int64)
        R._chk(R.lib().tpgref_knn_f32(R._f(a), R._f(b), R._i64(l1), R._i64(l2), B, P1, P2, D, K,
                                      C.c_float(-1.0 if r2 is None else r2), R._f(dist), R._i64(idx)),
               "knn")
        return _t(dist), _t(idx)

    def chamfer_fwd(self, src, tgt):
        return tuple(_t(x) for x in R.chamfer_fwd(_np(src), _np(tgt)))

    def chamfer_bwd(self, src, tgt, i1, i2, g1, g2):
        return tuple(_t(x) for x in R.chamfer_bwd(_np(src), _np(tgt), _np(i1), _np(i2), _np(g1), _np(g2)))

    def fps(self, xyz, m, start=None, skip_origin=True):
        if start is None and skip_origin:
            return _t(R.fps(_np(xyz), m))
        return _t(R.fps_start(_np(xyz), m, None if start is None else _np(start), skip_origin))

    def gather_fwd(self, feat, idx):
        return _t(R.gather_fwd(_np(feat), _np(idx)))

    def gather_bwd(self, gout, idx, N):
        return _t(R.gather_bwd(_np(gout), _np(idx), N))

    def gather_rows_fwd(self, rows, idx):
        return _t(R.gather_rows_fwd(_np(rows), _np(idx)))

    def gather_rows_bwd(self, gout, idx, N):
        return _t(R.gather_rows_bwd(_np(gout), _np(idx), N))

    def ball_query(self, radius, nsample, xyz, new_xyz):
        return _t(R.ball_query(radius, nsample, _np(xyz), _np(new_xyz)))

    def group_fwd(self, feat, idx):
        return _t(R.group_fwd(_np(feat), _np(idx)))

    def group_bwd(self, gout, idx, N):
        return _t(R.group_bwd(_np(gout), _np(idx), N))

    def three_nn(self, unknown, known):
        d2, idx = R.three_nn(_np(unknown), _np(known))
        return _t(d2), _t(idx)

    def three_interp_fwd(self, feat, idx, w):
        return _t(R.three_interp_fwd(_np(feat), _np(idx), _np(w)))

    def three_interp_bwd(self, gout, idx, w, m):
        return _t(R.three_interp_bwd(_np(gout), _np(idx), _np(w), m))


    # ---- row combine (fp32 on the CPU path) ----------------------------------------------
    def rowcombine_fwd(self, U, QE, idx, mode, slope, out_dtype):
        out = R.rowcombine_fwd(_np(U.float()), None if QE is None else _np(QE.float()), _np(idx), mode, slope)
        return _t(out).to(out_dtype)

    def rowcombine_bwd(self, gout, idx, E, mode, N, slope, in_dtype):
        gU, gQE = R.rowcombine_bwd(_np(gout.float()), _np(idx), None if E is None else _np(E.float()), mode,
                                   N, slope)
        return _t(gU).to(in_dtype), (None if gQE is None else _t(gQE).to(in_dtype))


    def rowcombine_edge_fwd(self, Y, idx, slope_a, slope_e, out_dtype):
        return _t(R.rowcombine_edge_fwd(_np(Y.float()), _np(idx), slope_a, slope_e)).to(out_dtype)

    def rowcombine_edge_bwd(self, gout, idx, Y, slope_a, slope_e, inverse=None):
        return _t(R.rowcombine_edge_bwd(_np(gout.float()), _np(idx), _np(Y.float()), slope_a, slope_e)).to(Y.dtype)

    def head_bn_act_fwd(self, h, gamma, beta, running_mean, running_var, nbt, momentum, eps, slope, mask):
        o = lambda t: None if t is None else _np(t.detach().float())
        y, mean, rstd, rm, rv = R.head_bn_act_fwd(o(h), o(gamma), o(beta), o(running_mean), o(running_var), momentum, eps,
                                                  slope, o(mask))
        with torch.no_grad():
            if running_mean is not None:
                running_mean.copy_(_t(rm))
            if running_var is not None:
                running_var.copy_(_t(rv))
            if nbt is not None:
                nbt += 1
        return _t(y), _t(mean), _t(rstd)

    def head_bn_act_bwd(self, gy, h, mean, rstd, gamma, beta, slope, mask, need_affine):
        o = lambda t: None if t is None else _np(t.detach().float())
        dh, dg, db = R.head_bn_act_bwd(o(gy), o(h), o(mean), o(rstd), o(gamma), o(beta), slope, o(mask))
        return _t(dh), (_t(dg) if need_affine else None), (_t(db) if need_affine else None)

    def cubic_interp(self, query, pos, field, cutoff):
        plain, pad, hits = R.cubic_interp(_np(query), _np(pos), _np(field), cutoff)
        return _t(plain), _t(pad), _t(hits)

    # ---- fused BatchNorm + act (+max) on rows ---------------------------------------------
    def rowbn_fwd(self, x, K, eps, momentum, training, running_mean, running_var, gamma, beta, slope, mean,
                  rstd, out_dtype, num_batches_tracked=None, nseg=1, mean_shift=None):
        """nseg segments = nseg consecutive calls on equal row blocks (include/tpgan_ops.h)."""
        Cc = x.shape[1]
        if not training and mean is None:            # identity statistics
            mean, rstd = torch.zeros(1, Cc), torch.ones(1, Cc)
        Ps = x.shape[0] // nseg
        ys, args = [], []
        for sg in range(nseg):
            xs = x[sg * Ps:(sg + 1) * Ps]
            ms = None if training else _np(mean.reshape(-1, Cc)[0])
            rs = None if training else _np(rstd.reshape(-1, Cc)[0])
            y, m, r, arg = R.rowbn_fwd(_np(xs.float()), K, eps, _np(gamma), _np(beta), slope, training, ms, rs)
            if training:
                mean.reshape(-1, Cc)[sg].copy_(_t(m))
                rstd.reshape(-1, Cc)[sg].copy_(_t(r))
                if num_batches_tracked is not None:
                    num_batches_tracked.add_(1)
                if running_mean is not None:
                    var = (1.0 / (_t(r).double() ** 2) - eps) * (Ps / max(Ps - 1, 1))
                    running_mean.mul_(1 - momentum).add_(momentum * (_t(m) + (0 if mean_shift is None else mean_shift)))
                    running_var.mul_(1 - momentum).add_((momentum * var).float())
            ys.append(_t(y))
            args.append(None if arg is None else _t(arg))
        y = torch.cat(ys, 0).to(out_dtype)
        return y, (None if args[0] is None else torch.cat(args, 0))

    def rowbn_bwd(self, gy, x, arg, K, training, mean, rstd, gamma, beta, slope, need_affine, y=None, nseg=1):
        Cc = x.shape[1]
        if mean is None:
            mean, rstd = torch.zeros(1, Cc), torch.ones(1, Cc)
        mean, rstd = mean.reshape(-1, Cc), rstd.reshape(-1, Cc)
        Ps, Gs = x.shape[0] // nseg, gy.shape[0] // nseg
        dxs, dg, db = [], 0.0, 0.0
        for sg in range(nseg):
            a = None if arg is None else _np(arg[sg * Gs:(sg + 1) * Gs])
            dx, g_, b_ = R.rowbn_bwd(_np(gy[sg * Gs:(sg + 1) * Gs].float()), _np(x[sg * Ps:(sg + 1) * Ps].float()), a, K,
                                     training, _np(mean[min(sg, mean.shape[0] - 1)]),
                                     _np(rstd[min(sg, rstd.shape[0] - 1)]), _np(gamma), _np(beta), slope)
            dxs.append(_t(dx))
            dg, db = dg + _t(g_), db + _t(b_)
        return torch.cat(dxs, 0).to(x.dtype), (dg if need_affine else None), (db if need_affine else None)


    # ---- fused spectral norm ------------------------------------------------------------
    def spectral_norm_fwd(self, W, u, v, iterate, eps):
        Wsn, u2, v2, sigma = R.spectral_norm_fwd(_np(W), _np(u), _np(v), iterate, eps)
        if iterate:
            u.copy_(_t(u2))
            v.copy_(_t(v2))
        return _t(Wsn), torch.tensor([float(sigma)])

    def spectral_norm_bwd(self, G, Wsn, u, v, sigma):
        return _t(R.spectral_norm_bwd(_np(G), _np(Wsn), _np(u), _np(v), float(sigma)))


    def spectral_norm_multi_fwd(self, Ws, us, vs, uses, iterate, eps):
        from tpgan_amd.ops import sn_multi_stride
        layout, goffs, dwoffs = [], [], []
        out_total = g_total = dw_total = 0
        for W, n in zip(Ws, uses):
            R_, Cn = W.shape
            st = sn_multi_stride(R_, Cn)
            layout.append((out_total, st))
            goffs.append(g_total)
            dwoffs.append(dw_total)
            out_total += st * n
            g_total += R_ * Cn * n
            dw_total += R_ * Cn
        flat = torch.zeros(out_total)
        for W, u, v, n, (off, st) in zip(Ws, us, vs, uses, layout):
            R_, Cn = W.shape
            for t in range(n):
                Wsn, u2, v2, sigma = R.spectral_norm_fwd(_np(W), _np(u), _np(v), iterate, eps)
                if iterate:
                    u.copy_(_t(u2))
                    v.copy_(_t(v2))
                o = off + t * st
                flat[o:o + R_ * Cn] = _t(Wsn).reshape(-1)
                flat[o + R_ * Cn:o + R_ * Cn + R_] = _t(u2)
                flat[o + R_ * Cn + R_:o + R_ * Cn + R_ + Cn] = _t(v2)
                flat[o + R_ * Cn + R_ + Cn] = float(sigma)
        plan = dict(layout=layout, goffs=goffs, dwoffs=dwoffs, dw_total=dw_total,
                    shapes=[tuple(W.shape) for W in Ws], uses=list(uses))
        return flat, plan

    def spectral_norm_multi_bwd(self, out, plan, gflat):
        dw = torch.zeros(plan["dw_total"])
        for (R_, Cn), n, (off, st), goff, dwoff in zip(plan["shapes"], plan["uses"], plan["layout"], plan["goffs"],
                                                       plan["dwoffs"]):
            acc = np.zeros((R_, Cn), np.float32)
            for t in range(n):
                o = off + t * st
                g = _np(gflat[goff + t * R_ * Cn: goff + (t + 1) * R_ * Cn]).reshape(R_, Cn)
                Wsn = _np(out[o:o + R_ * Cn]).reshape(R_, Cn)
                u = _np(out[o + R_ * Cn:o + R_ * Cn + R_])
                v = _np(out[o + R_ * Cn + R_:o + R_ * Cn + R_ + Cn])
                acc += R.spectral_norm_bwd(g, Wsn, u, v, float(out[o + R_ * Cn + R_ + Cn]))
            dw[dwoff:dwoff + R_ * Cn] = _t(acc).reshape(-1)
        return dw


def install():
    import tpgan_amd.ops as ops
    ops.register_backend("cpu", OracleBackend())


def uninstall():
    import tpgan_amd.ops as ops
    ops.unregister_backend("cpu")
