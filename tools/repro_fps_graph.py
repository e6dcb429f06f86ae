"""Minimal reproducer hunt: FPS inside a captured graph with other branches (round 3).

    python tools/repro_fps_graph.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import ops  # noqa: E402
from tpgan_amd.synthetic import fluid_clip  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    be = ops.backend_for(torch.zeros(1, device=dev))
    reps = int(os.environ.get("REPS", "60"))
    _, hb = fluid_clip(20, 16384, 4, 1, seed=3, device=dev)
    xb = hb[0].contiguous()
    c0 = be.fps(xb, 1024)
    x1 = torch.gather(xb, 1, c0.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()     # (20,1024,3), FPS ordered
    ref1 = be.fps(x1, 256)
    _, hs = fluid_clip(4, 16384, 4, 1, seed=5, device=dev)
    xs = hs[0].contiguous()
    nz = torch.randn(2048, 2048, device=dev)
    feat = torch.randn(24, 512, 64, device=dev)
    torch.cuda.synchronize()
    variants = {
        "fps20x1024 alone": dict(),
        "+ long fps(4,16384) on a 2nd branch": dict(long=True),
        "+ matmuls on the main branch": dict(mm=True),
        "+ feature-space knn on the main branch": dict(knn=True),
        "+ all": dict(long=True, mm=True, knn=True),
    }
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for name, v in variants.items():
        out = {}

        def body():
            main_s = torch.cuda.current_stream(dev)
            sa.wait_stream(main_s)
            sb.wait_stream(main_s)
            with torch.cuda.stream(sa):
                out["short"] = [be.fps(x1, 256) for _ in range(3)]
            if v.get("long"):
                with torch.cuda.stream(sb):
                    out["long"] = be.fps(xs, 1024)
            if v.get("mm"):
                y = nz
                for _ in range(6):
                    y = (y @ nz) * 1e-3
                out["y"] = y
            if v.get("knn"):
                out["knn"] = [be.knn(feat, feat, None, None, 20, None) for _ in range(6)]
            main_s.wait_stream(sa)
            main_s.wait_stream(sb)
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        bad_replay = bad_eager = 0
        for r in range(reps):
            g.replay()
            torch.cuda.synchronize()
            bad_replay += sum(not torch.equal(o, ref1) for o in out["short"])
        keep = out
        out = {}
        for r in range(reps):
            with torch.cuda.stream(s):
                body()
            torch.cuda.synchronize()
            bad_eager += sum(not torch.equal(o, ref1) for o in out["short"])
        print(f"{name}: wrong short-FPS results: replay {bad_replay} of {3 * reps}, eager {bad_eager} of {3 * reps}")
        del keep


if __name__ == "__main__":
    main()
