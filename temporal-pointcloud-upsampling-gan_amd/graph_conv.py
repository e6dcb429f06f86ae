"""Graph convolutions of the generator, computed on channels-last rows.

Host-side mirror of the reference's `gcn_lib/pointnet/gcn.py` (knn_query :13-22,
Dilated/DilatedKnnGraph :48-93, build_shared_mlp :96-120, conv_bn_layer :123-147,
EdgeConv :150-212, IDGCNLayer :215-279).  Module/attribute names, parameter shapes and
Sequential indices are identical, so state dicts interchange with the reference
(SURVEY.md Appendix B).

What is organised differently (MI355X-first, results equal up to fp32 rounding and pinned
by tests/golden): features travel as rows (B,N,C) instead of (B,C,N[,k]) planes, every 1x1
convolution is a GEMM on rows (`F.linear`, hipBLASLt), and the first layer of each grouped
MLP is applied BEFORE the gather -- a 1x1 convolution commutes with a gather:

    reference   node_affine(group(f)) + edge_affine(group(f) - f_i)         on N*k positions
    here        A = lrelu(Wn f), E = We f on N points; h = A[idx] + lrelu(E[idx] - E_i)

so the (B,C,N,k) grouped input is never built and the first-layer FLOPs drop k-fold; the
gather itself is one coalesced row-copy kernel (ops.row_combine, csrc/rowgather.hip).
Layers with batch/instance norm or spectral norm keep the reference order
(`_forward_planes`); the generator never builds those.
"""
import contextlib
import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import ops

_NORMS = ("batch", "ins", "none")

# Order of operations.  True (default): channels-last rows, first MLP layer before the gather
# (the MI355X path).  False: the reference's own order (group -> conv on grouped positions),
# bit-compatible with the reference's arithmetic; used to hold golden parity at rounding level
# where discrete neighbour decisions / training-mode BatchNorm amplify 1e-7 differences.
_ROWS_FIRST = [True]


def rows_first():
    return _ROWS_FIRST[0]


# the EdgeConv MLPs through ops.mlp_tail_plain (off: hipBLASLt GEMMs + separate activations, for A/B runs)
FUSED_EDGE_TAILS = [True]


@contextlib.contextmanager
def reference_order(enabled=True):
    """Run generator and discriminators in the reference's order of operations."""
    prev, _ROWS_FIRST[0] = _ROWS_FIRST[0], not enabled
    try:
        yield
    finally:
        _ROWS_FIRST[0] = prev


def knn_query(k, xyz1, xyz2=None):
    """(dist, idx int64) of the k nearest xyz2 rows per xyz1 row (gcn.py:13-22)."""
    if xyz2 is None:
        xyz2 = xyz1
    dist, idx = ops.neighbour_search(xyz1, xyz2, k)
    return ops.attach_dist_grad(xyz1, xyz2, dist, idx), idx


def radius_query(radius, sample, xyz1, xyz2=None, knn_padding=True):
    """gcn.py:25-45 (defined there, never called): <=sample hits within radius, -1 slots
    optionally replaced by the same-slot kNN index."""
    if xyz2 is None:
        xyz2 = xyz1
    dist, idx = ops.neighbour_search(xyz1, xyz2, sample, r=radius)
    if knn_padding:
        _, kidx = ops.neighbour_search(xyz1, xyz2, sample)
        idx = torch.where(idx == -1, kidx, idx)
    return dist, idx


def _norm_name(bn, insn):
    if insn and bn:
        raise Exception("Cant use batch normalization and instance normalization at the same time")
    return "batch" if bn else ("ins" if insn else "none")


def _conv1x1(cin, cout, norm, sn):
    # The reference's bias flag is inverted (gcn.py:98,102-106,125,128-132): a conv followed
    # by batch/instance norm HAS a bias, a bare conv (norm == 'none') has NONE.
    if norm not in _NORMS:
        raise Exception(f"Unsupported normalization: {norm}")
    conv = nn.Conv2d(cin, cout, kernel_size=1, bias=norm in ("batch", "ins"))
    layers = [spectral_norm(conv) if sn else conv]
    if norm == "batch":
        layers.append(nn.BatchNorm2d(cout))
    elif norm == "ins":
        layers.append(nn.InstanceNorm2d(cout))
    return layers


def build_shared_mlp(mlp_spec: List[int], norm: str = "batch", sn: bool = False):
    layers = []
    for i in range(1, len(mlp_spec)):
        layers += _conv1x1(mlp_spec[i - 1], mlp_spec[i], norm, sn)
        layers.append(nn.LeakyReLU(0.2))
    return nn.Sequential(*layers)


def conv_bn_layer(in_feat, out_feat, act=False, norm="batch", sn=False):
    layers = _conv1x1(in_feat, out_feat, norm, sn)
    if act:
        layers.append(nn.LeakyReLU(0.2))
    return nn.Sequential(*layers)


# Low-precision shadow copies of parameters.  Under autocast every use of a weight is its own
# cast launch (~2.5 us of kernel, ~5 us of dependent-launch latency in a hipGraph: ~35 of them in
# one generator forward).  A step that owns its parameters (gan_step_graph.GraphedFluidStep keeps
# them in one flat buffer) casts the whole buffer with ONE launch per step and registers the views
# here; `shadow_cast` then finds them by address.  Nothing is registered outside such a step.
_SHADOW = {"on": False, "dtype": None, "map": {}}


def shadow_cast(t, dtype):
    """t.to(dtype) -- or, inside a step that registered shadows, the copy it already made."""
    if _SHADOW["on"] and dtype == _SHADOW["dtype"] and t.dtype != dtype and t.is_contiguous():
        hit = _SHADOW["map"].get((t.data_ptr(), t.numel()))
        if hit is not None:
            return hit.view(t.shape)
    return t.to(dtype)


class shadows:
    """with shadows(flat_buffers, parameters, dtype): ... -- inside, `shadow_cast(p, dtype)` of a
    registered parameter (or a contiguous same-size view of it) returns a view of ONE buffer that
    `refresh()` fills with a single cast launch.  The caller refreshes after every change of the
    parameters and before their first use."""

    def __init__(self, flats, params, dtype):
        self.dtype = dtype
        self.flats = [f for f in flats if f.is_floating_point() and f.dtype != dtype]
        self.copies = [torch.empty_like(f, dtype=dtype) for f in self.flats]
        self.map = {}
        for p in params:
            for f, c in zip(self.flats, self.copies):
                off = (p.data_ptr() - f.data_ptr()) // f.element_size()
                if p.dtype == f.dtype and 0 <= off and off + p.numel() <= f.numel() and p.is_contiguous():
                    self.map[(p.data_ptr(), p.numel())] = c[off:off + p.numel()]

    def refresh(self):
        for f, c in zip(self.flats, self.copies):
            c.copy_(f)

    def __enter__(self):
        self._saved = dict(_SHADOW)
        _SHADOW.update(on=True, dtype=self.dtype, map=self.map)
        return self

    def __exit__(self, *exc):
        _SHADOW.update(self._saved)
        return False


# Weight gradients beside the chain.  Nothing downstream of a backward pass waits for dW before the optimizer,
# but on ONE stream its launches (split-K product + sum of the partials, bias sum) sit between two data-gradient
# launches of a latency-bound chain.  A step that owns a spare stream registers it here for the stream its
# backward runs on (gan_step_graph: an index-plan stream, idle by then); the weight-gradient launches go there --
# parallel leaves of the captured graph -- and the step joins that stream before its optimizers run.
_WGRAD_SIDE = {}


@contextlib.contextmanager
def wgrad_side_stream(main, side):
    """Inside: `_TallLinear*` backward passes running on stream `main` compute dW / db on `side`.  The caller makes
    `main` (or whatever reads the gradients) wait for `side` afterwards."""
    if side is None:
        yield
        return
    _WGRAD_SIDE[main.cuda_stream] = side
    try:
        yield
    finally:
        _WGRAD_SIDE.pop(main.cuda_stream, None)


def _wgrad_ctx(*reads):
    """Context for the weight-gradient launches of a backward pass: the registered side stream (ordered behind
    everything the current stream has issued, `reads` kept alive for it) or nothing."""
    if not _WGRAD_SIDE or not reads[0].is_cuda:
        return contextlib.nullcontext()
    cur = torch.cuda.current_stream(reads[0].device)
    side = _WGRAD_SIDE.get(cur.cuda_stream)
    if side is None:
        return contextlib.nullcontext()
    side.wait_stream(cur)
    for t in reads:
        t.record_stream(side)
    return torch.cuda.stream(side)


class _TallLinear(torch.autograd.Function):
    """y = x @ W^T for TALL x (P rows >> C): same forward GEMM as F.linear, but the weight
    gradient dW = gy^T x -- a (Cout x Cin) output with K = P up to 2.6e5 -- is computed as a
    split-K batched GEMM.  hipBLASLt maps the plain form to (Cout/64)*(Cin/64) = 4..16
    workgroups on a 256-CU part (380 us per call measured); S slices of the row axis give
    S*4..16 workgroups and finish in a few tens of us.  Partial products are summed in fp32."""

    @staticmethod
    def forward(ctx, x, w, bias, dtype):
        xd = x.to(dtype)
        wd = shadow_cast(w, dtype)
        ctx.save_for_backward(xd, wd)
        ctx.w_dtype, ctx.x_dtype = w.dtype, x.dtype
        ctx.b_dtype = None if bias is None else bias.dtype
        P, cin, cout, e = xd.shape[0], xd.shape[1], wd.shape[0], xd.element_size()
        if bias is None:
            return ops.timed("gemm_fwd", e * P * (cin + cout), 2 * P * cin * cout, xd, lambda: xd @ wd.t())
        return ops.timed("gemm_fwd", e * P * (cin + cout), 2 * P * cin * cout, xd,
                         lambda: torch.addmm(shadow_cast(bias, dtype), xd, wd.t()))  # bias in the GEMM epilogue

    @staticmethod
    def backward(ctx, gy):
        xd, wd = ctx.saved_tensors
        gy = gy.to(xd.dtype)
        dx = dw = db = None
        P, cin, cout, e = xd.shape[0], xd.shape[1], wd.shape[0], xd.element_size()
        if ctx.needs_input_grad[0]:
            dx = ops.timed("gemm_dgrad", e * P * (cin + cout), 2 * P * cin * cout, gy, lambda: gy @ wd).to(ctx.x_dtype)
        with _wgrad_ctx(gy, xd):
            if ctx.needs_input_grad[1]:
                dw = _tall_wgrad(gy, xd).to(ctx.w_dtype)
            if ctx.b_dtype is not None and ctx.needs_input_grad[2]:
                db = gy.sum(0, dtype=torch.float32).to(ctx.b_dtype)
        return dx, dw, db, None


def _tall_wgrad(gy, xd):
    """dW (Cout,Cin) fp32 = gy (P,Cout)^T xd (P,Cin) as a split-K batched GEMM (see _TallLinear)."""
    P, cout, cin, e = gy.shape[0], gy.shape[1], xd.shape[1], xd.element_size()
    S = _split_k(P, cout, cin)
    if S > 1:
        rows = P // S
        head = S * rows
        dw = ops.timed("gemm_wgrad", e * P * (cin + cout), 2 * P * cin * cout, gy,
                       lambda: _mm_f32(torch.bmm, gy[:head].view(S, rows, -1).transpose(1, 2),
                                       xd[:head].view(S, rows, -1))).sum(0)
        if head < P:
            dw = dw + _mm_f32(torch.mm, gy[head:].t(), xd[head:])
        return dw
    return _mm_f32(torch.mm, gy.t(), xd)


_F32_OUT = [None]     # does this torch build take out_dtype on mm / bmm?  (probed once, on the device)


def _mm_f32(fn, a, b):
    """fn(a, b) accumulated AND stored in fp32 (no bf16 rounding of the split-K partials, no
    separate cast kernel) where the build offers out_dtype; else the product cast to fp32."""
    if a.dtype == torch.float32:
        return fn(a, b)
    if _F32_OUT[0] is None:
        try:
            fn(a, b, out_dtype=torch.float32)
            _F32_OUT[0] = True
        except (TypeError, RuntimeError, NotImplementedError):
            _F32_OUT[0] = False
    return fn(a, b, out_dtype=torch.float32) if _F32_OUT[0] else fn(a, b).float()


class _TallLinearSeg(torch.autograd.Function):
    """nseg independent tall products in one batched GEMM: rows x (nseg*P, Cin) are nseg equal
    consecutive blocks, block s is multiplied by its own weight w[s] (Cout, Cin) -- the same
    module called nseg times (T frames, fake / real batch) with the successive spectrally
    normalised weights of those calls.  No bias: the caller folds it into the BatchNorm that
    follows (ops.row_bn_act, mean_shift).  Weight gradients per block by split-K like _TallLinear."""

    @staticmethod
    def forward(ctx, x, w, dtype):
        nseg, cout, cin = w.shape
        P = x.shape[0] // nseg
        xd = x.to(dtype).view(nseg, P, cin)
        wd = w.to(dtype)
        ctx.save_for_backward(xd, wd)
        ctx.w_dtype, ctx.x_dtype = w.dtype, x.dtype
        return ops.timed("gemm_fwd", xd.element_size() * nseg * P * (cin + cout), 2 * nseg * P * cin * cout, xd,
                         lambda: torch.bmm(xd, wd.transpose(1, 2))).view(nseg * P, cout)

    @staticmethod
    def backward(ctx, gy):
        xd, wd = ctx.saved_tensors
        nseg, P, cin = xd.shape
        cout = wd.shape[1]
        gy = gy.to(xd.dtype).view(nseg, P, cout)
        dx = dw = None
        e = xd.element_size()
        if ctx.needs_input_grad[0]:
            dx = ops.timed("gemm_dgrad", e * nseg * P * (cin + cout), 2 * nseg * P * cin * cout, gy,
                           lambda: torch.bmm(gy, wd)).view(nseg * P, cin).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            with _wgrad_ctx(gy, xd):
                S = _split_k(P, cout, cin)
                rows = P // S
                head = S * rows
                part = ops.timed("gemm_wgrad", e * nseg * P * (cin + cout), 2 * nseg * P * cin * cout, gy,
                                 lambda: _mm_f32(torch.bmm, gy[:, :head].reshape(nseg * S, rows, cout).transpose(1, 2),
                                                 xd[:, :head].reshape(nseg * S, rows, cin)))
                dw = part.view(nseg, S, cout, cin).sum(1) if S > 1 else part.view(nseg, cout, cin)
                if head < P:
                    dw = dw + _mm_f32(torch.bmm, gy[:, head:].transpose(1, 2), xd[:, head:])
                dw = dw.to(ctx.w_dtype)
        return dx, dw, None


def _stacked_weight(we, wn):
    """[We; Wn] as ONE (2H, Cin) matrix: a view when the two matrices sit back to back in one allocation (a step that
    keeps its parameters in a flat buffer has them in registration order, edge_affine right before node_affine), else
    their concatenation (one launch)."""
    if (we.shape == wn.shape and we.dtype == wn.dtype and we.is_contiguous() and wn.is_contiguous()
            and wn.data_ptr() == we.data_ptr() + we.numel() * we.element_size()
            and we.untyped_storage().data_ptr() == wn.untyped_storage().data_ptr()):
        return we.as_strided((2 * we.shape[0], we.shape[1]), (we.shape[1], 1), we.storage_offset())
    return torch.cat([we, wn], 0)


EDGE_FRONT = [os.environ.get("TPGAN_EDGE_FRONT", "1") != "0"]     # EdgeConv front end on one product (off: two GEMMs + LeakyReLU + ops.row_combine, for A/B runs)


class _EdgeFront(torch.autograd.Function):
    """h[b,n,k,:] = lrelu(Wn f[idx]) + lrelu(We f[idx] - We f[n]) of an EdgeConv (gcn.py:176-180,207-210) from ONE
    product Y = f [We; Wn]^T (csrc/rowgather.hip, tpg_rowcombine_edge_*): one GEMM, one data-gradient GEMM and one
    split-K weight gradient instead of two each, the node term's LeakyReLU and its derivative inside the gather
    kernels, no gradient sum of the shared input -- 7 launches fewer per EdgeConv and step on a latency-bound chain."""

    @staticmethod
    def forward(ctx, x, we, wn, idx, slope_a, slope_e, out_dtype, inverse):
        B, N, cin = x.shape
        H = we.shape[0]
        P = B * N
        wc = _stacked_weight(we.detach(), wn.detach())
        xr = x.reshape(P, cin)
        Y = ops.timed("gemm_fwd", 4 * P * (cin + 2 * H), 4 * P * cin * H, xr, lambda: xr @ wc.t()).view(B, N, 2 * H)
        h = ops.backend_for(x).rowcombine_edge_fwd(Y, idx, slope_a, slope_e, out_dtype)
        ctx.save_for_backward(xr, wc, Y, idx)
        ctx.inverse, ctx.slopes, ctx.shape = inverse, (slope_a, slope_e), (B, N, cin, H)
        return h

    @staticmethod
    def backward(ctx, gh):
        xr, wc, Y, idx = ctx.saved_tensors
        B, N, cin, H = ctx.shape
        P = B * N
        gh = gh.contiguous()
        if gh.dtype not in (torch.float32, torch.bfloat16):
            gh = gh.float()
        kw = {} if ctx.inverse is None else {"inverse": ctx.inverse}
        gY = ops.backend_for(gh).rowcombine_edge_bwd(gh, idx, Y, ctx.slopes[0], ctx.slopes[1], **kw).view(P, 2 * H)
        dx = dwe = dwn = None
        if ctx.needs_input_grad[0]:
            dx = ops.timed("gemm_dgrad", 4 * P * (cin + 2 * H), 4 * P * cin * H, gY, lambda: gY @ wc).view(B, N, cin)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            with _wgrad_ctx(gY, xr):
                dw = _tall_wgrad(gY, xr)
            dwe, dwn = dw[:H], dw[H:]
        return dx, dwe, dwn, None, None, None, None, None


def edge_front(x, we, wn, idx, slope_a, slope_e, out_dtype):
    """x (B,N,Cin) fp32 rows, we / wn (H,Cin) the bare edge / node convolutions, idx (B,N,K) int32 -> (B,N,K,H)."""
    inv = getattr(idx, "_tpg_inverse", None)
    inverse = inv[1:] if inv is not None and inv[0] == x.shape[1] else None
    return _EdgeFront.apply(x.contiguous(), we, wn, idx, float(slope_a), float(slope_e), out_dtype, inverse)


def _act(y, slope):
    if slope == 1.0:
        return y
    return torch.relu(y) if slope == 0.0 else F.leaky_relu(y, slope)


def _hand_dtype(x):
    """Output dtype of a hand-written row-linear launch on x: the autocast dtype when autocast is on (what the
    library GEMM under autocast would return), else x's own."""
    if x.is_cuda and torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype('cuda')
        return dt if dt in (torch.float32, torch.bfloat16) else None
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else None


def rows_matmul_seg(x, w, slope=1.0):
    """x (nseg*P, Cin) rows in nseg equal blocks, w (nseg, Cout, Cin): lrelu_slope(block s times w[s]^T)."""
    hd = _hand_dtype(x)
    if hd is not None and x.dtype in (torch.float32, torch.bfloat16) and ops.row_linear_supported(x, w, w.shape[0]):
        with torch.autocast(device_type=x.device.type, enabled=False):       # csrc/rowlinear.hip: one launch, fp32 products
            return ops.row_linear(x, w, None, slope, w.shape[0], hd)
    dtype = torch.get_autocast_dtype('cuda') if (x.is_cuda and torch.is_autocast_enabled()) else x.dtype
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        dtype = torch.float32
    with torch.autocast(device_type=x.device.type, enabled=False):
        return _act(_TallLinearSeg.apply(x, w, dtype), slope)


def _split_k(P, cout, cin):
    """Slices of the row axis for the weight-gradient GEMM: aim at >= 512 workgroups of 64x64
    output tiles, keep >= 1024 rows per slice."""
    tiles = max(1, (cout + 63) // 64) * max(1, (cin + 63) // 64)
    S = min(max(1, 512 // tiles), P // 1024)
    return max(1, S)


TALL_ROWS = 8192   # row counts from which the split-K weight gradient pays


def rows_matmul(x, w, bias=None, slope=1.0):
    """lrelu_slope(x (...,Cin) @ w (Cout,Cin)^T [+ bias]) on rows (slope 1 = no activation, 0 = ReLU): the library GEMM
    (tall inputs with the split-K backward) + the activation; with ops.ROW_LINEAR on, ONE hand-written launch
    (ops.row_linear, csrc/rowlinear.hip: bias, activation and the output's rounding in the epilogue) -- correct, but
    measured slower than the library at every shape of the step, hence off by default (see ops.ROW_LINEAR)."""
    hd = _hand_dtype(x)
    if hd is not None and x.dtype in (torch.float32, torch.bfloat16) and ops.row_linear_supported(x, w):
        with torch.autocast(device_type=x.device.type, enabled=False):
            return ops.row_linear(x, w, bias, slope, 1, hd)
    return _act(_rows_matmul_lib(x, w, bias), slope)


def _rows_matmul_lib(x, w, bias=None):
    lead = x.shape[:-1]
    P = x.numel() // x.shape[-1] if x.numel() else 0
    if P >= TALL_ROWS and x.is_cuda and torch.is_grad_enabled() and w.requires_grad:
        dtype = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype
        if dtype in (torch.float32, torch.bfloat16, torch.float16):
            with torch.autocast(device_type="cuda", enabled=False):
                y = _TallLinear.apply(x.reshape(P, x.shape[-1]), w, bias, dtype)
            return y.view(*lead, w.shape[0])
    return F.linear(x, w, bias)


def rows_linear(conv, x, slope=1.0):
    """Apply a bare 1x1 Conv2d (weight (Cout,Cin,1,1), optional bias) [+ LeakyReLU(slope)] to rows (...,Cin)."""
    return rows_matmul(x, conv.weight.view(conv.out_channels, -1), conv.bias, slope)


def rows_seq(seq, x):
    """Run a Sequential of bare 1x1 convs and LeakyReLUs on rows (norm == 'none' only); a conv and the LeakyReLU
    after it are one launch."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, nn.LeakyReLU) and 0.0 <= nxt.negative_slope <= 1.0:
                x = rows_linear(m, x, float(nxt.negative_slope))
                i += 1
            else:
                x = rows_linear(m, x)
        else:
            x = m(x)
        i += 1
    return x


def amp_dtype(x):
    """dtype the big grouped tensors should be stored in: the autocast dtype if autocast is
    active for x's device, else x's own dtype."""
    if x.is_cuda and torch.is_autocast_enabled():
        return torch.get_autocast_dtype('cuda')
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


def no_autocast(x):
    """Context in which the small pre-gather GEMMs run in fp32: their outputs are subtracted
    from each other after the gather (U[idx] - Q, E[idx] - E_i), so they must not be rounded
    to bf16 first (the reference subtracts in fp32 before its first conv)."""
    return torch.autocast(device_type=x.device.type, enabled=False)


class Dilated(nn.Module):
    """Every `dilation`-th entry of a k-NN list (gcn.py:48-72)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation, self.stochastic, self.epsilon = k, dilation, stochastic, epsilon

    def forward(self, edge_index):
        if self.stochastic and self.training and torch.rand(1) < self.epsilon:
            num = self.k * self.dilation
            randnum = torch.randperm(num)[:self.k]
            return edge_index[:, :, randnum]
        return edge_index[:, :, ::self.dilation]


class DilatedKnnGraph(nn.Module):
    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation = k, dilation
        self._dilated = Dilated(k, dilation, stochastic, epsilon)

    def forward(self, x):
        """x (B,N,C) -> (B,N,k//dilation) int64."""
        _, idx = ops.neighbour_search(x, x, self.k)
        return self._dilated(idx)


def _agg(name, y, dim):
    if name == "sum":
        return y.sum(dim)
    if name == "max":
        return y.max(dim)[0]
    if name == "min":
        return y.min(dim)[0]
    return y.mean(dim)


class EdgeConv(nn.Module):
    """kNN -> edge features -> shared MLP -> aggregate over k (gcn.py:150-212)."""

    def __init__(self, in_feat, out_feat, k=9, dilation=1, mlp_layer=True, aggregate="max",
                 bn=True, insn=False, sn=False, **kwargs):
        super().__init__()
        self.norm = _norm_name(bn, insn)
        self.k = k // dilation
        self.dilated_knn_graph = DilatedKnnGraph(k, dilation, **kwargs)
        half = out_feat // 2
        self.edge_affine = conv_bn_layer(in_feat, half, act=True, norm=self.norm, sn=sn)
        self.node_affine = conv_bn_layer(in_feat, half, act=True, norm=self.norm, sn=sn)
        self.mlp_layer = mlp_layer
        if mlp_layer:
            self.mlp = build_shared_mlp([half, half, out_feat], norm=self.norm, sn=sn)
        else:
            self.mlp = conv_bn_layer(half, out_feat, norm=self.norm, sn=sn, act=False)
        if aggregate not in ("sum", "max", "min", "mean"):
            raise Exception(f"Unsupported aggregation mode {aggregate}")
        self.aggregate = aggregate
        self._rows_ok = self.norm == "none" and not sn

    # ---- channels-last fast path ---------------------------------------------------------
    def forward_rows(self, x, pos=None, knn_idx=None):
        """x (B,N,C) rows [, pos (B,N,3) to search in] -> (B,N,C_out).
        knn_idx: an already searched neighbour list of the same points with at least this
        layer's k entries (the first k of a longer kNN list ARE the k-NN list: canonical
        ascending (dist, idx) order); the layer's own dilation is applied to it."""
        if not (self._rows_ok and rows_first()):
            return self._forward_planes(x.transpose(1, 2), pos).squeeze(-1).transpose(1, 2)
        x = x.contiguous()
        if knn_idx is not None:
            g = self.dilated_knn_graph
            idx = g._dilated(knn_idx[:, :, :g.k]).to(torch.int32).contiguous()
        else:
            idx = self.dilated_knn_graph(pos if pos is not None else x).to(torch.int32).contiguous()
        na, ea = self.node_affine[0], self.edge_affine[0]
        H = na.out_channels
        if (EDGE_FRONT[0] and na.bias is None and ea.bias is None and H == ea.out_channels
                and H % (4 if amp_dtype(x) == torch.float32 else 8) == 0):
            with no_autocast(x):                                         # one product for both terms (_EdgeFront)
                h = edge_front(x.float(), ea.weight.view(H, -1), na.weight.view(H, -1), idx, 0.2, 0.2, amp_dtype(x))
        else:
            with no_autocast(x):
                xf = x.float()
                A = rows_linear(na, xf, 0.2)                             # (B,N,H), LeakyReLU in the epilogue
                E = rows_linear(ea, xf)
            h = ops.row_combine(A, E, idx, ops.ROW_EDGE, slope=0.2, out_dtype=amp_dtype(x))   # (B,N,k,H)
        if self.mlp_layer:
            mods = list(self.mlp)
            C_out = mods[-2].out_channels if isinstance(mods[-2], nn.Conv2d) else 0
            fused = self._fused_tail(mods, h)
            if fused is not None:
                return fused
            if (self.aggregate == "max" and isinstance(mods[-1], nn.LeakyReLU) and C_out % 8 == 0
                    and 0 < C_out <= 1024 and h.shape[2] <= 256):
                # last [LeakyReLU -> max over k] fused into one pass (ops.row_act_max)
                y = rows_seq(mods[:-1], h)                              # (B,N,k,C_out), pre-activation
                B, N, k, _ = y.shape
                out = ops.row_act_max(y.reshape(B * N * k, C_out), mods[-1].negative_slope, k)
                return out.view(B, N, C_out)
            return _agg(self.aggregate, rows_seq(self.mlp, h), 2)
        if self.aggregate in ("sum", "mean"):                           # linear commutes with sum
            return rows_seq(self.mlp, _agg(self.aggregate, h, 2))
        return _agg(self.aggregate, rows_seq(self.mlp, h), 2)

    def _fused_tail(self, mods, h):
        """[conv, LeakyReLU]* -> max over the k neighbours of h (B,N,k,H) on the fused MFMA kernels
        (ops.mlp_tail_plain, csrc/mlp_fused.hip) where they take the shapes: bf16 rows on the GPU, supported
        channel pairs, bare convs.  -> (B,N,C_out) or None."""
        if not (FUSED_EDGE_TAILS[0] and self.aggregate == "max" and h.is_cuda):
            return None
        if (len(mods) == 4 and all(isinstance(m, nn.Conv2d) and m.bias is None for m in mods[0::2])
                and all(isinstance(m, nn.LeakyReLU) and 0.0 <= m.negative_slope <= 1.0 for m in mods[1::2])
                and ops.small_tail_supported(h, (h.shape[-1], mods[0].out_channels, mods[2].out_channels), h.shape[2])
                and mods[0].in_channels == h.shape[-1] and mods[2].in_channels == mods[0].out_channels):
            # the IDGCN blocks' 16 -> 16 -> 32 tails: one launch each way on the vector ALUs (csrc/mlp_small.hip)
            B, N, k, H = h.shape
            out = ops.small_tail(h.reshape(B * N * k, H), mods[0].weight.view(mods[0].out_channels, -1),
                                 mods[2].weight.view(mods[2].out_channels, -1), mods[1].negative_slope,
                                 mods[3].negative_slope, k)
            return out.view(B, N, mods[2].out_channels)
        if h.dtype != torch.bfloat16:
            return None
        convs, slopes = [], [1.0]
        for i in range(0, len(mods), 2):
            conv, act = mods[i], mods[i + 1] if i + 1 < len(mods) else None
            if not (isinstance(conv, nn.Conv2d) and conv.bias is None and isinstance(act, nn.LeakyReLU)
                    and 0.0 <= act.negative_slope <= 1.0):
                return None
            convs.append(conv)
            slopes.append(float(act.negative_slope))
        B, N, k, H = h.shape
        chans = [H] + [c.out_channels for c in convs]
        if not ops.mlp_tail_supported(h, chans, k) or any(c.in_channels != a for c, a in zip(convs, chans[:-1])):
            return None
        Ws = [c.weight.view(c.out_channels, -1).float() for c in convs]
        out = ops.mlp_tail_plain(h.view(B * N * k, H), Ws, slopes, k)
        return out.view(B, N, chans[-1])

    # ---- reference order, any norm -------------------------------------------------------
    def _forward_planes(self, feat, pos=None):
        feat = feat.contiguous().float()                                # (B,C,N)
        search_in = pos if pos is not None else feat.transpose(1, 2)
        knn_idx = self.dilated_knn_graph(search_in).to(torch.int32).contiguous()
        grouped = ops.grouping_operation(feat, knn_idx)                 # (B,C,N,k)
        out = self.node_affine(grouped) + self.edge_affine(grouped - feat.unsqueeze(-1))
        return _agg(self.aggregate, self.mlp(out), -1).unsqueeze(-1)

    def forward(self, feat, pos=None):
        """Reference signature: feat (B,C,N[,1]) -> (B,C_out,N,1)."""
        if feat.dim() == 4 and feat.shape[-1] == 1:
            feat = feat.squeeze(-1)
        if not (self._rows_ok and rows_first()):
            return self._forward_planes(feat, pos)
        return self.forward_rows(feat.transpose(1, 2), pos).transpose(1, 2).unsqueeze(-1)


class IDGCNLayer(nn.Module):
    """Inception-DenseGCN block (gcn.py:215-279)."""

    def __init__(self, in_feats, out_feats, bn=True, insn=False, ln=False, sn=False, residual=True):
        super().__init__()
        self.norm = _norm_name(bn, insn)
        q = in_feats // 4
        self.btn = conv_bn_layer(in_feats, q, act=False, norm=self.norm, sn=sn)
        self.GCN1 = EdgeConv(q, q, k=20, dilation=1, aggregate="max", mlp_layer=True, bn=bn, insn=insn, sn=sn)
        self.GCN2 = EdgeConv(q, q, k=20, dilation=2, aggregate="max", mlp_layer=True, bn=bn, insn=insn, sn=sn)
        self.decoder = conv_bn_layer(q * 3, out_feats, act=True, norm=self.norm, sn=sn)
        self.use_layernorm = ln
        if ln:
            self.layernorm = nn.LayerNorm([out_feats])
        self.residual = residual
        if residual:
            self.skip_layer = conv_bn_layer(in_feats, out_feats, act=False, norm=self.norm, sn=sn)
        self._rows_ok = self.norm == "none" and not sn

    def forward_rows(self, x):
        """x (B,N,C) -> (B,N,C_out)."""
        if not (self._rows_ok and rows_first()):
            return self._forward_planes(x.transpose(1, 2).unsqueeze(-1)).squeeze(-1).transpose(1, 2)
        skip = rows_seq(self.skip_layer, x) if self.residual else None
        low = rows_seq(self.btn, x).contiguous()                        # (B,N,C/4)
        # ONE search serves the three neighbour lists built on `low` (gcn.py:253-262 searches three
        # times): the local 9-NN list and the two EdgeConvs' k-NN lists are prefixes of the longest
        kmax = max(9, self.GCN1.dilated_knn_graph.k, self.GCN2.dilated_knn_graph.k)
        # (the search and both EdgeConvs work on fp32 rows: ONE upcast of `low` instead of one per use)
        low_f = low.float()
        _, idx_all = ops.neighbour_search(low_f, low_f, kmax)
        idx = idx_all[:, :, :9]
        local = ops.row_combine(low, None, idx.to(torch.int32).contiguous(), ops.ROW_GATHER)   # (B,N,9,C/4)
        B, N, k, q = local.shape
        if q % 8 == 0 and low.dtype in (torch.float32, torch.bfloat16):
            # max over the 9 neighbours with a one-byte arg-max and a one-pass backward
            local_max = ops.row_act_max(local.view(B * N * k, q), 1.0, k).view(B, N, q)
        else:
            local_max = local.max(2)[0]
        out = torch.cat([local_max, self.GCN1.forward_rows(low_f, knn_idx=idx_all),
                         self.GCN2.forward_rows(low_f, knn_idx=idx_all)], dim=-1)
        out = rows_seq(self.decoder, out)
        if self.use_layernorm:
            out = self.layernorm(out)
        return out + skip if self.residual else out

    def _forward_planes(self, feature):                                 # (B,C,N,1), reference order
        skip = self.skip_layer(feature) if self.residual else None
        low = self.btn(feature).squeeze(-1).contiguous().float()
        _, idx = ops.neighbour_search(low.transpose(1, 2), low.transpose(1, 2), 9)
        local = ops.grouping_operation(low, idx.to(torch.int32).contiguous())
        out = self.decoder(torch.cat([torch.max(local, dim=-1, keepdim=True)[0], self.GCN1(low),
                                      self.GCN2(low)], dim=1))
        if self.use_layernorm:
            B, C, N, _ = out.shape
            out = self.layernorm(out.squeeze(-1).permute(0, 2, 1).reshape(-1, C))
            out = out.reshape(B, N, C).permute(0, 2, 1).unsqueeze(-1).contiguous()
        return out + skip if self.residual else out

    def forward(self, feature):
        """Reference signature: (B,C,N,1) -> (B,C_out,N,1)."""
        if not (self._rows_ok and rows_first()):
            return self._forward_planes(feature)
        return self.forward_rows(feature.squeeze(-1).transpose(1, 2)).transpose(1, 2).unsqueeze(-1)
