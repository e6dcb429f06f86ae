"""Every launch of the hand-written kernels in one step, by (kernel, algorithmic bytes): count, time and
bandwidth of each distinct shape, each launch timed ALONE (device synchronised before it, so no other
stream shares the HBM).  Shows which shapes sit far from the roofline.  GPU box.

    python tools/kernel_shapes.py [name prefix ...]        # default: rowbn_ rowcombine_
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                             # noqa: E402

prefixes = tuple(sys.argv[1:]) or ("rowbn_", "rowcombine_")
sys.argv = [sys.argv[0]]
import bench
from tpgan_amd.synthetic import fluid_clip  # noqa: E402
from tpgan_amd import ops                                                # noqa: E402
from tpgan_amd.gan_step_graph import GraphedFluidStep                    # noqa: E402


class AloneTimer(ops.OpTimer):
    def record(self, name, nbytes, stream_of, launch):
        torch.cuda.synchronize(stream_of.device)
        return super().record((name, int(nbytes)), nbytes, stream_of, launch)


def main():
    import numpy as np
    dev = torch.device("cuda", 0)
    np.random.seed(0)
    G, Ds, Dt, opts = bench.build(dev, capturable=True)
    clips = [fluid_clip(8, 4096, 8, 3, seed=s, device=dev) for s in range(2)]
    step = GraphedFluidStep(G, Ds, Dt, opts, bench.OPT, clips[0][0], clips[0][1], 1.0, torch.bfloat16, None)
    step._load(*clips[1])
    step._run_eager(True)
    torch.cuda.synchronize()
    timer = AloneTimer()
    ops.set_timer(timer)
    reps = 3
    for _ in range(reps):
        step._run_eager(True)
    torch.cuda.synchronize()
    ops.set_timer(None)
    rows = []
    for (name, nbytes), recs in timer.pending.items():
        if not name.startswith(prefixes):
            continue
        us = sorted(1e3 * a.elapsed_time(b) for a, b, _ in recs)
        med = us[len(us) // 2]
        rows.append((len(recs) / reps * med, name, nbytes, len(recs) / reps, med, nbytes / med / 1e3))
    rows.sort(reverse=True)
    tot = collections.Counter()
    print("%-24s %12s %6s %9s %9s %9s" % ("kernel", "MB/launch", "n", "median us", "GB/s", "ms/step"))
    for total, name, nbytes, n, med, gbps in rows:
        tot[name] += total
        print("%-24s %12.2f %6.1f %9.1f %9.0f %9.3f" % (name, nbytes / 1e6, n, med, gbps, total / 1e3))
    print({k: round(v / 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
