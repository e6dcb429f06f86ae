"""First kernel whose result changes from run to run of the SAME step body (timing dependence), with optional
background load to shake the schedule.

Every HipBackend call (the hand-written kernels) records, on its own stream, a bitwise checksum of every tensor it was
given and returned (inputs AFTER the call: what library GEMMs / PyTorch ops in between produced shows up as a changed
input of the next record).  The step body of a fresh stepper is run REPS times from identical state; records are
compared with run 0 per (stream, sequence number).

    python tools/race_trace.py cfg2 8 bf16 [noise] [replay-free: body only]
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import configs, ops  # noqa: E402


def _sum_bits(t):
    t = t.detach()
    if not t.is_cuda or t.numel() == 0:
        return None
    t = t.contiguous()
    if t.dtype in (torch.float32, torch.int32):
        v = t.view(torch.int32)
    elif t.dtype in (torch.bfloat16, torch.float16, torch.int16):
        v = t.view(torch.int16)
    elif t.dtype in (torch.int64, torch.float64):
        v = t.view(torch.int64)
    elif t.dtype in (torch.uint8, torch.bool, torch.int8):
        v = t.view(torch.uint8)
    else:
        return None
    return v.sum(dtype=torch.int64)


def _tensors(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _tensors(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors(o, out)
    return out


def main():
    name, batch, amp_s = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    noise = "noise" in sys.argv[4:]
    reps = int(os.environ.get("REPS", "6"))
    amp = torch.bfloat16 if amp_s == "bf16" else None
    torch.backends.cudnn.enabled = False
    dev = torch.device("cuda", 0)
    A = configs.build_models(name, dev, seed=5, capturable=True)
    for m in list(A[1].modules()) + list(A[2].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clip = configs.make_clip(name, batch=batch, seed=1, device=dev)
    be = ops.backend_for(torch.zeros(1, device=dev))
    records = []          # (stream id, name, [checksum tensors])
    clones = []           # fps calls: (stream id, [input, output]) kept whole
    active = [False]
    for k in dir(be):
        f = getattr(be, k)
        if k.startswith("_") or not callable(f) or k in ("lib",):
            continue

        def make(k, f):
            def g(*a, **kw):
                pre = None
                if active[0] and k == "fps":
                    pre = (_sum_bits(a[0]), a[0].detach().clone())      # the input as a kernel BEFORE the launch sees it
                out = f(*a, **kw)
                if active[0]:
                    sid = torch.cuda.current_stream(dev).cuda_stream
                    ts = _tensors((a, kw, out), [])
                    if k.startswith("spectral_norm_multi"):
                        ts = [t for t in ts if t.dtype != torch.int64 and t.dtype != torch.uint8]   # (descriptor tables hold addresses)
                    records.append((sid, k, [s for s in map(_sum_bits, ts) if s is not None]))
                    if k == "fps":
                        clones.append((sid, [t.detach().clone() for t in ts] + [pre[1]]))
                return out
            return g
        setattr(be, k, make(k, f))
    noise_stream = torch.cuda.Stream(dev)
    nz = torch.randn(3072, 3072, device=dev)
    runs = []
    for r in range(reps):
        M = copy.deepcopy(A[:3])
        M = (*M, tuple(torch.optim.Adam(m.parameters(), lr=g.param_groups[0]["lr"], capturable=True)
                       for m, g in zip((M[0], M[2], M[1]), A[3])))
        st = configs.graphed_step(name, M, clip, amp_dtype=amp)
        torch.cuda.synchronize()
        if noise:          # a varying amount of unrelated work on another stream while the step runs
            with torch.cuda.stream(noise_stream):
                for _ in range(40 + 37 * r):
                    nz2 = nz @ nz
        configs.seed_host_rng(3)
        records.clear()
        clones.clear()
        active[0] = True
        losses = st(clip[0], clip[1], 12, launch_eagerly=True)
        active[0] = False
        torch.cuda.synchronize()
        # per stream, in order: streams get different ids per stepper -> rank them by first appearance
        order = {}
        for sid, _, _ in records:
            order.setdefault(sid, len(order))
        flat = [s for _, _, sums in records for s in sums]
        vals = torch.stack(flat).cpu().tolist()
        per, i = {}, 0
        for sid, k, sums in records:
            per.setdefault(order[sid], []).append((k, vals[i:i + len(sums)]))
            i += len(sums)
        fps_calls = {}
        for sid, ts in clones:
            fps_calls.setdefault(order[sid], []).append([t.cpu() for t in ts])
        runs.append((losses, per, fps_calls))
        print(f"run {r}: tempo_D {losses['tempo_D_loss']:.7f} spatial_D {losses['spatial_D_loss']:.7f} "
              f"tempo_G {losses['tempo_G_loss']:.7f} spatial_G {losses['spatial_G_loss']:.7f}; records per stream "
              f"{[len(v) for v in per.values()]}")
        del st
    base = runs[0][1]
    clean = True
    for r in range(1, reps):
        per = runs[r][1]
        for s in base:
            a, b = base[s], per.get(s, [])
            if len(a) != len(b):
                print(f"run {r} stream {s}: {len(b)} records vs {len(a)}")
            for i, (ra, rb) in enumerate(zip(a, b)):
                if ra != rb:
                    clean = False
                    which = [j for j, (x, y) in enumerate(zip(ra[1], rb[1])) if x != y]
                    print(f"run {r} stream {s}: first difference at record {i}/{len(a)} {ra[0]} (tensors {which} of {len(ra[1])});"
                          f" preceding: {[x[0] for x in a[max(0, i - 4):i]]}; following: {[x[0] for x in a[i + 1:i + 3]]}")
                    break
    print("box clean: every record of every run equals run 0" if clean else "differences found")
    # the FPS calls themselves, element by element
    for r in range(1, reps):
        for s, calls in runs[0][2].items():
            for c, (a, b) in enumerate(zip(calls, runs[r][2].get(s, []))):
                same_in = torch.equal(a[0], b[0])
                for tag, run in (("run 0", a), (f"run {r}", b)):
                    if not torch.equal(run[0], run[2]):
                        nbad = int((run[0] != run[2]).any(-1).sum())
                        print(f"{tag} stream {s} fps call {c}: the input read BEFORE the launch differs from the input read AFTER it "
                              f"in {nbad} points -> the launch did not wait for its producer")
                if not torch.equal(a[1], b[1]):
                    ia, ib = a[1], b[1]
                    clouds = [int(i) for i in torch.nonzero((ia != ib).any(1)).flatten()]
                    print(f"run {r} stream {s} fps call {c}: input {tuple(a[0].shape)} equal={same_in}; output differs in clouds {clouds}")
                    for cl in clouds[:2]:
                        first = int(torch.nonzero(ia[cl] != ib[cl]).flatten()[0])
                        nd = int((ia[cl] != ib[cl]).sum())
                        print(f"    cloud {cl}: first differing pick {first} of {ia.shape[1]} ({nd} picks differ): run0 {ia[cl, first:first + 6].tolist()} "
                              f"run{r} {ib[cl, first:first + 6].tolist()}; same set of picks: {sorted(ia[cl].tolist()) == sorted(ib[cl].tolist())}")
                        x = a[0][cl]
                        # what SHOULD the pick at `first` be?  recompute the running distances on the CPU from the agreed prefix
                        pts = x.double()
                        d = torch.full((pts.shape[0],), 1e10, dtype=torch.float64)
                        for j in range(first):
                            d = torch.minimum(d, ((pts - pts[ia[cl, j]]) ** 2).sum(1))
                        elig = (pts ** 2).sum(1) > 1e-3
                        d[~elig] = -1
                        top = torch.topk(d, 3)
                        print(f"    fp64 recomputation: best candidates {top.indices.tolist()} with d2 {top.values.tolist()}")


if __name__ == "__main__":
    main()
