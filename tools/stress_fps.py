"""Does a search kernel give the same answer when other streams keep the chip busy?  (round 3: GPUTEST lottery)

    python tools/stress_fps.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import ops  # noqa: E402
from tpgan_amd.synthetic import fluid_clip  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    be = ops.backend_for(torch.zeros(1, device=dev))
    iters = int(os.environ.get("ITERS", "200"))
    s_main, s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    nz = torch.randn(2048, 2048, device=dev)
    big = torch.randn(64 * 1024 * 1024 // 4, device=dev)
    # the step's own situation: a long FPS of many large clouds on one stream, short FPS calls on another
    _, hb = fluid_clip(20, 16384, 4, 1, seed=3, device=dev)
    xb = hb[0].contiguous()
    ref_big = be.fps(xb, 1024)
    _, hs = fluid_clip(4, 16384, 4, 1, seed=5, device=dev)
    xs = torch.gather(hs[0], 1, be.fps(hs[0].contiguous(), 1024).long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    ref_small = be.fps(xs, 512)
    torch.cuda.synchronize()
    bad_small = bad_big = 0
    outs = []
    for it in range(iters // 4):
        with torch.cuda.stream(s_a):
            big_out = be.fps(xb, 1024)
        with torch.cuda.stream(s_b):
            y = nz @ nz
        with torch.cuda.stream(s_main):
            outs = [be.fps(xs, 512) for _ in range(6)]
        torch.cuda.synchronize()
        bad_small += sum(not torch.equal(o, ref_small) for o in outs)
        bad_big += not torch.equal(big_out, ref_big)
    print(f"fps (4,1024)->512 beside fps (20,16384)->1024: {bad_small} of {6 * (iters // 4)} short calls wrong, "
          f"{bad_big} of {iters // 4} long calls wrong")
    for B, N, m in ((4, 1024, 512), (8, 4096, 1024), (4, 16384, 1024), (24, 1024, 256), (8, 512, 128)):
        _, high = fluid_clip(B, N, 4, 1, seed=7, device=dev)
        x = high[0].contiguous()
        ref = be.fps(x, m)
        ctr = torch.gather(x, 1, ref.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        ref_bq = be.ball_query(0.1, 32, x, ctr)
        torch.cuda.synchronize()
        bad_f = bad_b = 0
        for it in range(iters):
            with torch.cuda.stream(s_a):
                for _ in range(1 + it % 3):
                    y = nz @ nz
            with torch.cuda.stream(s_b):
                z = big * 1.0001
            with torch.cuda.stream(s_main):
                got = be.fps(x, m)
                gb = be.ball_query(0.1, 32, x, ctr)
                eq_f = torch.equal(got, ref)
                eq_b = torch.equal(gb, ref_bq)
            bad_f += not eq_f
            bad_b += not eq_b
        torch.cuda.synchronize()
        print(f"fps ({B},{N})->{m}: {bad_f} of {iters} runs differ from the quiet result; ball_query: {bad_b}")


if __name__ == "__main__":
    main()
