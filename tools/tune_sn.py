"""Spectral-norm forward of a discriminator's weight set: the split kernel against the one-workgroup kernel, per
number of chained uses.  GPU box.

    python tools/tune_sn.py [cfg2|cfg4]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import tpgan_amd  # noqa: F401
from tpgan_amd import ops

SHAPES = {
    "cfg2": [(1, 64), (64, 6), (64, 256), (128, 64), (128, 131), (128, 256), (256, 128), (256, 128), (256, 256),
             (256, 256), (256, 256), (256, 256), (256, 259), (256, 515), (256, 515)],
    "cfg4": [(1, 64), (64, 6), (64, 64), (64, 256), (128, 64), (128, 131)] + [(128, 256)] * 6 + [(256, 128)] * 7
            + [(256, 256)] * 2 + [(256, 259), (256, 512)] + [(256, 515)] * 7 + [(512, 256)],
}


def main(name):
    dev = torch.device("cuda", 0)
    hip = ops.backend_for(torch.zeros(1, device=dev))
    torch.manual_seed(0)
    Ws = [torch.randn(s, device=dev) * 0.1 for s in SHAPES[name]]
    for uses in (1, 2, 3, 6, 16):
        row = []
        for split in (True, False):
            ops.SN_SPLIT[0] = split
            hip._sn_plans.clear()
            us = [torch.nn.functional.normalize(torch.randn(s[0], device=dev), dim=0) for s in SHAPES[name]]
            vs = [torch.nn.functional.normalize(torch.randn(s[1], device=dev), dim=0) for s in SHAPES[name]]
            n = [uses] * len(Ws)
            for _ in range(3):
                hip.spectral_norm_multi_fwd(Ws, us, vs, n, True, 1e-12)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                hip.spectral_norm_multi_fwd(Ws, us, vs, n, True, 1e-12)
            b.record()
            torch.cuda.synchronize()
            row.append(a.elapsed_time(b) / 50 * 1e3)
        print(f"{name}: {len(Ws)} weights x {uses:2d} uses   split {row[0]:7.1f} us   one workgroup {row[1]:7.1f} us  (launch-to-launch, "
              f"memset node included)", flush=True)
    ops.SN_SPLIT[0] = True


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
