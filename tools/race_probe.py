"""Is the step body reproducible run to run, and if not, which concurrency makes it timing dependent?

Builds the graphed stepper N times from the same state / seeds and runs ONE step (body launched kernel by kernel, or
the replay) per build, under a variant that removes one source of concurrency at a time:

    base       the step as shipped (index plans on two side streams, the two discriminator updates on two branches)
    nobranch   both discriminator updates on the main stream
    noside     both index-plan streams = the main stream
    serial     nobranch + noside
    rocblas / hipblaslt   torch.backends.cuda.preferred_blas_library(...)

Prints the distinct loss dictionaries seen and a checksum of every network's state after the step.

    python tools/race_probe.py cfg5shard 4 fp32 body base nobranch noside serial
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import configs  # noqa: E402


def checksum(m):
    tot = 0
    for v in m.state_dict().values():
        v = v.detach().contiguous()
        if v.dtype == torch.float32:
            tot += int(v.view(torch.int32).sum(dtype=torch.int64))
        elif v.dtype == torch.int64:
            tot += int(v.sum())
    return tot & 0xffffffffffff


def main():
    name, batch, amp_s, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    variants = sys.argv[5:] or ["base"]
    reps = int(os.environ.get("REPS", "5"))
    amp = torch.bfloat16 if amp_s == "bf16" else None
    torch.backends.cudnn.enabled = False
    dev = torch.device("cuda", 0)
    A = configs.build_models(name, dev, seed=5, capturable=True)
    for m in list(A[1].modules()) + list(A[2].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clip = configs.make_clip(name, batch=batch, seed=1, device=dev)
    for variant in variants:
        if variant in ("rocblas", "hipblaslt"):
            torch.backends.cuda.preferred_blas_library(variant)
        seen = {}
        plans = []
        for r in range(reps):
            M = copy.deepcopy(A[:3])
            M = (*M, tuple(torch.optim.Adam(m.parameters(), lr=g.param_groups[0]["lr"], capturable=True)
                           for m, g in zip((M[0], M[2], M[1]), A[3])))
            st = configs.graphed_step(name, M, clip, amp_dtype=amp)
            main_s = torch.cuda.current_stream(dev)
            if variant in ("nobranch", "serial"):
                st.branch = st.branch2 = main_s
            if variant in ("noside", "serial"):
                st.sides = [main_s, main_s]
            torch.cuda.synchronize()
            configs.seed_host_rng(3)
            losses = st(clip[0], clip[1], 12, launch_eagerly=(mode == "body"))
            torch.cuda.synchronize()
            from tpgan_amd.set_abstraction import _plan_tensors
            keep = {}
            for kname in ("fake_s", "true_s", "fakes", "trues", "plan_true_s", "plan_true_t", "plan_s", "plan_t"):
                tot = 0
                for t in _plan_tensors(st._keep.get(kname)):
                    t = t.detach().contiguous()
                    v = t.view(torch.int32) if t.dtype in (torch.float32, torch.int32) else t.view(torch.int64)
                    tot += int(v.sum(dtype=torch.int64))
                keep[kname] = tot & 0xffffffffff
            key = (tuple(sorted(losses.items())), tuple(checksum(m) for m in M[:3]), tuple(sorted(keep.items())))
            plans.append({kname: [t.detach().cpu() for t in _plan_tensors(st._keep.get(kname))]
                          for kname in ("plan_true_s", "plan_true_t", "trues", "true_s")})
            seen[key] = seen.get(key, 0) + 1
            del st
        print(f"{name} batch {batch} {amp_s} {mode} variant {variant}: {len(seen)} distinct outcomes in {reps} runs")
        for (losses, sums, keep), n in seen.items():
            d = dict(losses)
            print(f"   x{n}: tempo_D {d['tempo_D_loss']:.7f} spatial_D {d['spatial_D_loss']:.7f} tempo_G {d['tempo_G_loss']:.7f} "
                  f"spatial_G {d['spatial_G_loss']:.7f} CD {d['Chamfer_distance_no_norm']:.5f}  state sums G/Ds/Dt {sums}")
            print("        kept:", dict(keep))
        # which plan tensor differs, and where (reference = the first run whose outcome is the most frequent one)
        import collections
        sig = [tuple(tuple(int(t.long().sum()) for t in p[k]) for k in sorted(p)) for p in plans]
        ref = sig.index(collections.Counter(sig).most_common(1)[0][0])
        for r, p in enumerate(plans):
            for kname in p:
                for ti, (a, b) in enumerate(zip(plans[ref][kname], p[kname])):
                    if not torch.equal(a, b):
                        ne = (a != b)
                        first = torch.nonzero(ne.reshape(-1))[0].item()
                        idx = list(torch.unravel_index(torch.tensor(first), a.shape))
                        print(f"   run {r}: {kname}[{ti}] shape {tuple(a.shape)} {a.dtype}: {int(ne.sum())} elements differ, first at "
                              f"{[int(i) for i in idx]}: ref {a.reshape(-1)[first:first + 6].tolist()} got {b.reshape(-1)[first:first + 6].tolist()}")


if __name__ == "__main__":
    main()
