"""Host time of every piece of GraphedFluidStep.__call__ (the GPU idles between the report's sync of
one step and the first kernel of the next).  GPU box.

    python tools/call_breakdown.py
"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402

sys.argv = [sys.argv[0]]
import bench
from tpgan_amd.synthetic import fluid_clip  # noqa: E402
from tpgan_amd.gan_step_graph import GraphedFluidStep                    # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    np.random.seed(0)
    G, Ds, Dt, opts = bench.build(dev, capturable=True)
    clips = [fluid_clip(8, 4096, 8, 3, seed=s, device=dev) for s in range(4)]
    step = GraphedFluidStep(G, Ds, Dt, opts, bench.OPT, clips[0][0], clips[0][1], 1.0, torch.bfloat16, None)
    for i in range(3):
        step(*clips[i % 4], 12)
    acc = collections.OrderedDict()

    def lap(name, t0):
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1
    n = 40
    torch.cuda.synchronize()
    wall0 = time.perf_counter()
    for i in range(n):                      # the body of __call__, piece by piece
        low, high = clips[i % 4]
        t = time.perf_counter()
        np_state, cpu_rng = np.random.get_state(), torch.get_rng_state()
        t = lap("numpy / torch CPU rng state", t)
        cuda_rng = torch.cuda.get_rng_state(dev)
        t = lap("cuda rng state", t)
        step._stage_host_draws(True)
        t = lap("host draws", t)
        step._load(low, high)
        t = lap("load clips (6 copies)", t)
        step._dev_f.copy_(step._host_f, non_blocking=True)
        step._dev_i.copy_(step._host_i, non_blocking=True)
        t = lap("staging H2D (2 copies)", t)
        for d, s_ in zip(step._snap, step._state):
            d.copy_(s_)
        t = lap("snapshot", t)
        for g, _ in step._graphs[True]:
            g.replay()
        t = lap("graph launch (host returns)", t)
        out = torch.cat([step.report, step.viol.reshape(1)]).cpu().tolist()
        t = lap("report: cat + sync + tolist", t)
        del np_state, cpu_rng, cuda_rng, out
    wall = time.perf_counter() - wall0
    print("%.3f ms per step in total" % (1e3 * wall / n))
    for k, v in acc.items():
        print("  %-34s %8.3f ms" % (k, 1e3 * v / n))


if __name__ == "__main__":
    main()
