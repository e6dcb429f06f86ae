"""One replayed step on the GPU timeline, from a rocprofv3 --kernel-trace CSV: how many kernels run side by
side over the step and what runs when.  CAVEAT, measured: under rocprofv3 --kernel-trace the dispatches of
the replayed graph are serialised -- 87 % of the step's wall time has exactly one kernel running and the step
lasts 21.4 ms = the SUM of its kernel durations (21.9 ms), against 14.2 ms unprofiled -- so this shows the
order of the work and its per-family shares, not the overlap the three branches reach without the profiler
(21.9 ms of kernels in 14.2 ms = 1.5 kernels side by side on average).

    python tools/timeline.py gpurun_out/.../NNNN_kernel_trace.csv
"""
import collections
import csv
import sys


def main(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    # steps are separated by the host sync: find gaps > 150 us with nothing running
    steps, cur, busy_until = [], [], 0
    for s, e, n, q in rows:
        if cur and s - busy_until > 150_000:
            steps.append(cur)
            cur = []
        cur.append((s, e, n, q))
        busy_until = max(busy_until, e)
    steps.append(cur)
    big = [st for st in steps if len(st) > 1200]
    print("%d launches, %d segments, %d look like full steps" % (len(rows), len(steps), len(big)))
    st = big[len(big) // 2]                          # a step from the middle of the run
    t0, t1 = st[0][0], max(e for _, e, _, _ in st)
    print("step: %d launches over %.2f ms on %d queues" % (len(st), (t1 - t0) / 1e6, len({q for *_, q in st})))
    ev = sorted([(s, 1) for s, *_ in st] + [(e, -1) for _, e, *_ in st])
    hist, level, last = collections.Counter(), 0, t0
    for t, d in ev:
        hist[level] += t - last
        level, last = level + d, t
    tot = t1 - t0
    print("kernels running side by side (share of the step's wall time):")
    for k in sorted(hist):
        print("   %d: %5.1f %%" % (k, 100 * hist[k] / tot))
    # coarse profile: 1-ms bins, busy fraction and the kernel family with most time in the bin
    nb = int((t1 - t0) / 1e6) + 1
    fam = [collections.Counter() for _ in range(nb)]
    for s, e, n, _ in st:
        key = ("GEMM" if n.startswith("Cijk") else "fps" if "fps_kernel" in n else "rowbn" if "rowbn" in n else
               "gather" if "rowcombine" in n or "rowsum" in n else "knn" if "knn" in n else
               "spectral" if "spectral" in n else "torch" if "at::native" in n else "other")
        b0, b1 = int((s - t0) / 1e6), int((e - t0) / 1e6)
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * 1_000_000), min(e, t0 + (b + 1) * 1_000_000)
            fam[b][key] += max(0, hi - lo)
    print("per millisecond: kernel time by family (us), sum / 1000 = average concurrency")
    for b, c in enumerate(fam):
        tot_b = sum(c.values())
        print("  %2d ms  %5.2fx  %s" % (b, tot_b / 1e6, "  ".join("%s %d" % (k, v / 1e3) for k, v in c.most_common(5))))


if __name__ == "__main__":
    main(sys.argv[1])
