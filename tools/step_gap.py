"""How much of a step's wall time is the host's: the full call (draws, staging, snapshot, replay, the
one host sync for the loss report) against back-to-back replays of the same graphs with one sync at
the end.  GPU box.

    python tools/step_gap.py
"""
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
    from tpgan_amd import configs
    clips = [configs.make_clip("cfg2", seed=s, device=dev) for s in range(4)]
    step = configs.graphed_step("cfg2", configs.build_models("cfg2", dev, capturable=True), clips[0], amp_dtype=torch.bfloat16)
    n = 40
    for i in range(3):
        step(*clips[i % 4], 12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(*clips[i % 4], 12)
    torch.cuda.synchronize()
    full = (time.perf_counter() - t0) / n
    graphs = step._graphs[True]
    t0 = time.perf_counter()
    for i in range(n):
        for g, _ in graphs:
            g.replay()
    launch = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    back = (time.perf_counter() - t0) / n
    one = []
    for i in range(5):                      # a single replay on an idle GPU: host time of the launch alone
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g, _ in graphs:
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        one.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
    print("single replay on an idle GPU: host returns after %.3f ms, GPU done after %.3f ms" % min(one))
    print("full call            %.3f ms/step" % (1e3 * full))
    print("back-to-back replays %.3f ms/step (host time to enqueue one replay: %.3f ms)" % (1e3 * back, 1e3 * launch))
    # the call's host-side pieces, timed alone
    for name, fn in (("host draws", lambda: step._stage_host_draws(True)), ("load clips", lambda: step._load(*clips[1])),
                     ("snapshot", lambda: [d.copy_(t) for d, t in zip(step._snap, step._state)])):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        host = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        print("%-20s %.3f ms of host time per step" % (name, 1e3 * host))


if __name__ == "__main__":
    main()
