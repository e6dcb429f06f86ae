"""Hand-written row-linear kernels (csrc/rowlinear.hip) against the library GEMM on the step's shapes: forward,
forward + backward, per call, replayed from a hipGraph (20 calls per replay).

    python tools/tune_rowlinear.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import graph_conv, ops  # noqa: E402

SHAPES = [(12288, 3, 64, 1, "f32"), (12288, 128, 128, 1, "bf16"), (12288, 128, 32, 1, "bf16"), (12288, 32, 16, 1, "f32"),
          (12288, 96, 128, 1, "bf16"), (12288, 256, 64, 1, "bf16"), (12288, 64, 128, 1, "f32"), (12288, 256, 128, 1, "bf16"),
          (196608, 6, 64, 6, "f32"), (49152, 131, 128, 6, "f32"), (12288, 259, 256, 2, "f32"), (8192, 515, 256, 4, "f32"),
          (8, 256, 256, 1, "f32")]


def bench(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def main():
    dev = "cuda"
    for P, Cin, Cout, nseg, dt in SHAPES:
        dtype = torch.float32 if dt == "f32" else torch.bfloat16
        x = torch.randn(P, Cin, device=dev).to(dtype).requires_grad_(True)
        W = (torch.randn(nseg, Cout, Cin, device=dev) / Cin ** 0.5).requires_grad_(True)
        gy = torch.randn(P, Cout, device=dev).to(dtype)
        res = {}
        for tag, on in (("hand", True), ("library", False)):
            ops.ROW_LINEAR[0] = on

            def fwd():
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dt == "bf16")):
                    if nseg > 1:
                        return graph_conv.rows_matmul_seg(x, W, 0.2)
                    return graph_conv.rows_matmul(x, W[0], None, 0.2)

            def both():
                y = fwd()
                torch.autograd.grad(y, [x, W], gy)
            res[tag] = (bench(lambda: fwd()), bench(both))
        ops.ROW_LINEAR[0] = False
        print(f"P={P:7d} {Cin:4d}->{Cout:4d} nseg={nseg} {dt}: forward hand {res['hand'][0]:7.1f} us  library {res['library'][0]:7.1f} us | "
              f"fwd+bwd hand {res['hand'][1]:7.1f} us  library {res['library'][1]:7.1f} us")


if __name__ == "__main__":
    main()
