"""Which op makes two identical runs differ?  (round 3: every end-to-end assertion needs a reproducible path.)

Runs the same computation twice from identical copies of the state under a TorchFunctionMode that records, for every
torch-level call, a bitwise checksum of its tensor inputs and outputs (on the device, one sync at the end), plus the
same for every HipBackend method (the hand-written kernels).  The first record whose INPUTS agree and whose OUTPUTS
differ names the non-reproducible op; inputs that differ first point at whatever produced them.

    python tools/find_nondeterminism.py cfg5shard 4 fp32 [Ds|Dt|step|body]
"""
import copy
import os
import sys

import torch
from torch.overrides import TorchFunctionMode

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


class Recorder(TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self.records = []          # (name, [input sums], [output sums])
        self.busy = False

    def note(self, name, ins, outs):
        self.busy = True           # (the checksums' own torch calls are not records)
        try:
            self.records.append((name, [s for s in map(_sum_bits, ins) if s is not None],
                                 [s for s in map(_sum_bits, outs) if s is not None]))
        finally:
            self.busy = False

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        if self.busy:
            return out
        name = getattr(func, "__qualname__", None) or getattr(func, "__name__", str(func))
        if any(w in name for w in ("empty", "data_ptr", "__get__", "numel", "size", "stride", "is_contiguous", "dim")):
            return out             # (uninitialised memory / metadata)
        # AFTER the call, inputs included (in-place ops, kernels writing into arguments)
        allt = _tensors((args, kwargs, out), [])
        if any(t.is_cuda for t in allt):
            self.note(name, [], allt)
        return out


def wrap_backend(rec):
    be = ops.backend_for(torch.zeros(1, device="cuda"))
    saved = {}
    for k in dir(be):
        f = getattr(be, k)
        if k.startswith("_") or not callable(f):
            continue

        def make(k, f):
            def g(*a, **kw):
                out = f(*a, **kw)
                rec.note("hip." + k, [], _tensors((a, kw, out), []))
                return out
            return g
        saved[k] = f
        setattr(be, k, make(k, f))
    return be, saved


def run_once(name, batch, amp, what, A, clip):
    M = copy.deepcopy(A[:3])
    dev = clip[0][0].device
    torch.manual_seed(11)
    stepper = None
    if what == "body":
        Mo = (*M, tuple(torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True) for m in (M[0], M[2], M[1])))
        stepper = configs.graphed_step(name, Mo, clip, amp_dtype=amp)
        torch.cuda.synchronize()
    rec = Recorder()
    be, saved = wrap_backend(rec)
    try:
        with rec:
            if what == "body":
                configs.seed_host_rng(3)
                print("   losses:", stepper(clip[0], clip[1], 12, launch_eagerly=True))
            elif what == "step":
                Mo = (*M, tuple(torch.optim.SGD(m.parameters(), lr=0.01) for m in (M[0], M[2], M[1])))
                configs.seed_host_rng(3)
                print("   losses:", configs.eager_step(name, Mo, clip, 12, amp_dtype=amp))
            else:
                D = M[1] if what == "Ds" else M[2]
                fake = [h + 0.003 * torch.randn_like(h) for h in clip[1]]
                with torch.autocast("cuda", dtype=amp) if amp is not None else torch.autocast("cuda", enabled=False):
                    if what == "Ds":
                        a, b = D.forward_passes([fake[1], clip[1][1]])
                    else:
                        a, b = D.forward_passes([fake, clip[1]], configs.opt_of(name).R)
                (((a.float() - 0.1) ** 2).mean() + ((b.float() - 1.0) ** 2).mean()).backward()
    finally:
        for k, f in saved.items():
            setattr(be, k, f)
    torch.cuda.synchronize()
    flat = [s for r in rec.records for s in r[1] + r[2]]
    vals = torch.stack(flat).cpu().tolist() if flat else []
    out, i = [], 0
    for nm, ins, outs in rec.records:
        out.append((nm, vals[i:i + len(ins)], vals[i + len(ins):i + len(ins) + len(outs)]))
        i += len(ins) + len(outs)
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5shard"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    amp = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else None
    what = sys.argv[4] if len(sys.argv) > 4 else "Ds"
    torch.backends.cudnn.enabled = False
    dev = torch.device("cuda", 0)
    A = configs.build_models(name, dev, seed=5)
    for m in list(A[1].modules()) + list(A[2].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    clip = configs.make_clip(name, batch=batch, seed=1, device=dev)
    runs = [run_once(name, batch, amp, what, A, clip) for _ in range(3)]
    print(f"{name} batch {batch} {'bf16' if amp else 'fp32'} {what}: {len(runs[0])} records per run")
    for first, other in ((1, 2),):                  # (run 0 warms one-time caches up)
        a, b = runs[first], runs[other]
        if len(a) != len(b):
            print(f"run {first} vs {other}: different record counts {len(a)} / {len(b)}")
        shown = 0
        for i, (ra, rb) in enumerate(zip(a, b)):
            if ra[0] != rb[0]:
                print(f"run {first} vs {other}: record {i} names differ {ra[0]} / {rb[0]}")
                break
            if ra[2] != rb[2] or ra[1] != rb[1]:
                which = [j for j, (x, y) in enumerate(zip(ra[2], rb[2])) if x != y]
                print(f"run {first} vs {other}: record {i} {ra[0]}: tensors {which} of {len(ra[2])} differ after the call"
                      f"   (previous records: {[r[0] for r in a[max(0, i - 3):i]]})")
                shown += 1
                if shown >= 8:
                    break
        if not shown:
            print(f"run {first} vs {other}: all {min(len(a), len(b))} records bitwise equal")


if __name__ == "__main__":
    main()
