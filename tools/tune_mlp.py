"""Fused MFMA tail-layer kernels alone, on the step's shapes: time per launch (hipGraph replay of 20
launches), algorithmic GB/s and TFLOP/s.  GPU box.

    python tools/tune_mlp.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import ops

dev = torch.device("cuda", 0)
hip = ops.backend_for(torch.zeros(1, device=dev))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3      # us


SHAPES = [  # (name, rows per segment, nseg, Cin, Cout)
    ("lvl0 64->128 x6", 8 * 1024 * 32, 6, 64, 128), ("lvl0 64->128 x3", 8 * 1024 * 32, 3, 64, 128),
    ("Ds lvl1 128->128 x2", 8 * 512 * 32, 2, 128, 128), ("Dt lvl1 128->256 x6", 8 * 256 * 32, 6, 128, 256),
    ("flow 256->128 x4", 8 * 256 * 32, 4, 256, 128), ("flow 256->256 x2", 8 * 256 * 32, 2, 256, 256)]
for name, P, nseg, Cin, Cout in SHAPES:
    x = torch.randn(P * nseg, Cin, device=dev).bfloat16()
    W = torch.randn(nseg, Cout, Cin, device=dev) / Cin ** 0.5
    ss = torch.rand(nseg, 2, Cin, device=dev)
    us = timeit(lambda: hip.mlp_fwd(x, ss, 0.01, W, nseg, 1e-5, 0.1, None, None, None, None, None, None))
    nbytes = 2 * P * nseg * (Cin + Cout)
    flops = 2 * P * nseg * Cin * Cout
    # the unfused sequence it replaces: BN apply, GEMM, BN statistics
    bn = torch.nn.BatchNorm1d(Cin).to(dev)
    def unfused():
        a = ops.row_bn_act(x, bn.weight, bn.bias, None, None, True, 0.1, 1e-5, 0.01, 0, nseg=nseg)
        y = torch.bmm(a.view(nseg, P, Cin), W.bfloat16().transpose(1, 2)).view(nseg * P, Cout)
        return hip.rowbn_fwd(y, 0, 1e-5, 0.1, True, None, None, None, None, 0.01, torch.empty(nseg, Cout, device=dev),
                             torch.empty(nseg, Cout, device=dev), torch.bfloat16, nseg=nseg, **{})
    try:
        us0 = timeit(unfused)
    except Exception as e:  # noqa: BLE001
        us0 = float("nan")
        print("unfused reference failed:", e)
    print(f"{name:22s} fused {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s  {flops / us / 1e6:7.1f} TFLOP/s   "
          f"| stats+apply+GEMM+stats unfused {us0:8.1f} us")


print()
for name, P, nseg, Cin, Cout in SHAPES:
    K = 32
    x_in = torch.randn(P * nseg, Cin, device=dev).bfloat16()
    x_out = torch.randn(P * nseg, Cout, device=dev).bfloat16()
    W = torch.randn(nseg, Cout, Cin, device=dev) / Cin ** 0.5
    cb = torch.rand(nseg, 4, Cout, device=dev)
    ci = torch.rand(nseg, 4, Cin, device=dev)
    ag = torch.randn(P * nseg // K, Cout, device=dev).bfloat16()
    arg = torch.randint(0, K, (P * nseg // K, Cout), device=dev, dtype=torch.uint8)
    gd = torch.randn(P * nseg, Cout, device=dev).bfloat16()
    for mode, g, a, k in (("max", ag, arg, K), ("dense", gd, None, 0)):
        us_d = timeit(lambda: hip.mlp_dgrad(x_out, g, a, k, cb, x_in, ci, 0.01, W, nseg, True))
        us_w = timeit(lambda: hip.mlp_wgrad(x_out, g, a, k, cb, x_in, ci, 0.01, nseg))
        bd = 2 * P * nseg * (Cout + 2 * Cin) + (2 * P * nseg * Cout if mode == "dense" else 0)
        bw = 2 * P * nseg * (Cout + Cin) + (2 * P * nseg * Cout if mode == "dense" else 0)
        print(f"{name:22s} {mode:5s} dgrad {us_d:8.1f} us {bd / us_d / 1e3:7.1f} GB/s | wgrad {us_w:8.1f} us {bw / us_w / 1e3:7.1f} GB/s")
    g1 = torch.randn(P * nseg, Cin, device=dev).bfloat16()
    c12 = torch.rand(nseg, 2, Cin, device=dev)
    us = timeit(lambda: hip.mlp_bn_bwd_apply(g1, x_in, ci, c12, nseg))
    us_r = timeit(lambda: hip.mlp_bn_bwd_apply(g1, x_in, ci, c12, nseg, K))
    print(f"{name:22s} bn_bwd_apply (C={Cin}) {us:8.1f} us {6 * P * nseg * Cin / us / 1e3:7.1f} GB/s | with the row sums over K = {K} "
          f"{us_r:8.1f} us {6 * P * nseg * Cin / us_r / 1e3:7.1f} GB/s")
