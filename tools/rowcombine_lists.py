"""Every row_combine backward of one step: shape, mode, time alone, and how skewed its inverted lists
are (mean / max entries per source row; a thread walks one row's whole list).  GPU box.

    python tools/rowcombine_lists.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402

sys.argv = [sys.argv[0]]
import bench
from tpgan_amd.synthetic import fluid_clip  # noqa: E402
from tpgan_amd import ops                                                # noqa: E402
from tpgan_amd.gan_step_graph import GraphedFluidStep                    # noqa: E402

rows = []
orig = ops.HipBackend.rowcombine_bwd


def spy(self, gout, idx, E, mode, N, slope, in_dtype, inverse=None):
    B, S, K, Cc = gout.shape
    offs, _ = inverse if inverse is not None else self.invert_index(idx, N)
    lens = (offs[:, 1:] - offs[:, :-1]).float()
    # padded slots of a ball query repeat the group's first entry
    pad = float((idx[:, :, 1:] == idx[:, :, :1]).float().mean()) if K > 1 else 0.0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(self, gout, idx, E, mode, N, slope, in_dtype, inverse)
    e1.record()
    torch.cuda.synchronize()
    mb = gout.element_size() * gout.numel() / 1e6
    rows.append((e0.elapsed_time(e1) * 1e3, mode, B, N, S, K, Cc, mb, float(lens.mean()), float(lens.max()),
                 float(lens.max(1).values.mean()), pad))
    return out


def main():
    dev = torch.device("cuda", 0)
    np.random.seed(0)
    G, Ds, Dt, opts = bench.build(dev, capturable=True)
    clips = [fluid_clip(8, 4096, 8, 3, seed=s, device=dev) for s in range(2)]
    step = GraphedFluidStep(G, Ds, Dt, opts, bench.OPT, clips[0][0], clips[0][1], 1.0, torch.bfloat16, None)
    step._load(*clips[1])
    step._run_eager(True)
    torch.cuda.synchronize()
    ops.HipBackend.rowcombine_bwd = spy
    step._run_eager(True)
    ops.HipBackend.rowcombine_bwd = orig
    print("%8s %4s %4s %6s %6s %4s %5s %9s %7s %7s %9s %6s %8s" % ("us", "mode", "B", "N", "S", "K", "C", "gout MB", "mean", "max", "mean max", "pad", "GB/s"))
    for r in sorted(rows, reverse=True):
        print("%8.1f %4d %4d %6d %6d %4d %5d %9.1f %7.1f %7.0f %9.1f %6.2f %8.0f" % (r + (r[7] / r[0] * 1e3,)))
    print("total %.3f ms" % (sum(r[0] for r in rows) / 1e3))


if __name__ == "__main__":
    main()
