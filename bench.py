"""Headline benchmark: GAN train-steps/s (full G+D adversarial step) on synthetic
4096-point x 3-frame fluid clips, batch 8 per GPU (BASELINE.json configs[1]; weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = `tempo_gan_step` semantics with generator AND both discriminators updated
(even iteration > 10, gate open, all-keep mask regime -- SURVEY.md section 8d).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N == 1 only):
  roofline      the hand-written kernel with the largest share of kernel time, timed live
                with HIP events on its launch stream inside a second run of the same steps;
  cpu_baseline  the same step function on the host cores through the oracle ops (kind
                "port": the reference has no CPU path for pointnet2_ops/FRNN), on a bounded
                sample (4 clips instead of 8), scaled to the metric's unit.
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import tpgan_amd  # noqa: E402
from tpgan_amd import ddp, ops  # noqa: E402
from tpgan_amd.gan_step import tempo_gan_step  # noqa: E402
from tpgan_amd.set_abstraction import FluidSpatialDis, FluidTempoDis  # noqa: E402
from tpgan_amd.srnet import SRNet  # noqa: E402
from tpgan_amd.synthetic import fluid_clip, force_all_keep  # noqa: E402

T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBPS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
OPT = Namespace(use_vel=False, in_node_feats=3, cutoff=0.025, R=0.10, w=0.5)


def build(device, seed=1, capturable=False):
    torch.manual_seed(seed)
    G = force_all_keep(SRNet(3, 128)).to(device)
    Ds = FluidSpatialDis().to(device)
    Dt = FluidTempoDis(3).to(device)
    lr = 3e-4
    # capturable for the hipGraph path; fused = torch's single-launch multi-tensor Adam (the
    # foreach form divides by per-parameter 0-dim step tensors one launch per parameter)
    kw = {"capturable": True, "fused": torch.device(device).type == "cuda"} if capturable else {}

    def adam(params, lr_):
        try:
            return torch.optim.Adam(params, lr=lr_, **kw)
        except (RuntimeError, ValueError):
            return torch.optim.Adam(params, lr=lr_, **{k: v for k, v in kw.items() if k != "fused"})
    opts = (adam(list(G.parameters()), lr), adam(list(Dt.parameters()), 0.33 * lr), adam(list(Ds.parameters()), 0.33 * lr))
    return G, Ds, Dt, opts


GRAPHED = None   # GraphedFluidStep when the hipGraph path is active
EAGER_BODY = False   # --eager-body: run the captured step body launch by launch (for PMC passes)
TRACE = bool(os.environ.get("TPGAN_BENCH_TRACE"))


def run_steps(models, clips, n, sync, amp_dtype, start=0):
    G, Ds, Dt, (og, ot, os_) = models
    out = None
    for i in range(n):
        low, high = clips[(start + i) % len(clips)]
        if GRAPHED is not None and low[0].is_cuda and EAGER_BODY:
            out = GRAPHED.run_body_eagerly(low, high)
        elif GRAPHED is not None and low[0].is_cuda:
            out = GRAPHED(low, high, 12)
        else:
            out = tempo_gan_step(G, Ds, Dt, low, None, high, None, 1.0, OPT, 12, og, ot, os_,
                                 sync=sync, amp_dtype=amp_dtype, force_gate=True)
        if TRACE:
            log("step %d: %s" % (start + i, {k: round(v, 4) for k, v in out.items()}))
    return out


def cpu_baseline(sample_batch, per_gpu_batch, n_hi):
    """Time ONE step of the same workload on the host cores (oracle ops + CPU PyTorch)."""
    # threads = the cores this process may actually run on (a GPU box hands each job a share)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("TPGAN_CPU_THREADS", "16"))))
    os.environ["OMP_NUM_THREADS"] = str(cores)      # read by the oracle's OpenMP runtime at load
    from oracle import torch_backend
    torch_backend.install()
    try:
        torch.set_num_threads(cores)
        np.random.seed(0)
        models = build("cpu")
        warm = [fluid_clip(2, 512, 8, 3, seed=7)]
        run_steps(models, warm, 1, None, None)                      # page in / thread pools
        clips = [fluid_clip(sample_batch, n_hi, 8, 3, seed=1234)]
        t0 = time.perf_counter()
        run_steps(models, clips, 1, None, None)
        dt = time.perf_counter() - t0
    finally:
        torch_backend.uninstall()
    return {"value": (sample_batch / per_gpu_batch) / dt, "unit": "steps/s", "cores": cores,
            "kind": "port",
            "sample": f"1 full G+D step on {sample_batch} clips of {n_hi} pts x3 frames "
                      f"({dt:.1f} s), scaled by {sample_batch}/{per_gpu_batch} to the batch-"
                      f"{per_gpu_batch} step; fp32; oracle C ops (OpenMP) + CPU PyTorch convs"}


def roofline_leg(models, clips, steps, sync, amp_dtype):
    """Per-kernel HIP-event timing needs individual launches: this leg always runs eagerly."""
    global GRAPHED
    timer = ops.OpTimer()
    ops.set_timer(timer)
    saved, GRAPHED = GRAPHED, None
    try:
        if saved is not None:            # the body the graphs replay, kernel by kernel
            for i in range(steps):
                saved.run_body_eagerly(*clips[i % len(clips)])
        else:
            run_steps(models, clips, steps, sync, amp_dtype)
        torch.cuda.synchronize()
    finally:
        ops.set_timer(None)
        GRAPHED = saved
    summ = timer.summary()
    # Streaming kernels are priced against the HBM roofline; furthest-point sampling and the
    # neighbour searches are chains of dependent rounds on cache-resident clouds (latency /
    # issue bound by construction) and are listed with their own figure instead.
    streaming = [k for k in summ if k.startswith(("rowbn_", "rowcombine_", "group_"))]
    name = max(streaming or list(summ), key=lambda k: summ[k]["total_ms"])
    dom = summ[name]
    roof = {"bound": "hbm", "kernel": name, "achieved": round(dom["gbps"], 2), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(dom["gbps"] / HBM_PEAK_GBPS, 6), "traffic": pmc_traffic(name),
            "avg_launch_us": round(dom["avg_us"], 2), "launches_per_step": dom["launches"] / steps,
            "algorithmic_bytes_per_launch": int(dom["bytes_per_launch"]),
            "selection": "largest total time among the HBM-streaming kernels; timed per launch with HIP "
                         "events while the captured step body runs eagerly (launch by launch)"}
    table = {k: {"launches_per_step": v["launches"] / steps, "ms_per_step": round(v["total_ms"] / steps, 4),
                 "avg_us": round(v["avg_us"], 2), "GBps": round(v["gbps"], 1)} for k, v in summ.items()}
    if "fps" in summ:
        table["fps"]["note"] = "npoint-1 dependent rounds per launch; hidden on a side stream in graph mode"
    return roof, table


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), with the
    gfx950 FETCH_SIZE x2 correction for wide coalesced reads; None when no PMC file is present."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh).get(kernel)
        return None if rec is None else rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--points", type=int, default=4096, help="high-res points per frame")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-extra", action="store_true", help="skip roofline and cpu_baseline legs")
    ap.add_argument("--cpu-sample-batch", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true",
                    help="run the step eagerly instead of replaying it from captured hipGraphs")
    ap.add_argument("--eager-body", action="store_true",
                    help="build the graphed step but run its body launch by launch (for rocprofv3 PMC "
                         "passes: per-kernel counters of the exact launches the graphs replay)")
    ap.add_argument("--miopen", action="store_true",
                    help="let PyTorch use MIOpen for conv/BN (first use JIT-compiles per shape: minutes)")
    args = ap.parse_args()
    # 1x1 convs and BatchNorm go through rocBLAS / native kernels; MIOpen would JIT-compile one
    # kernel per new shape on a fresh box, which swamps any short run.
    torch.backends.cudnn.enabled = bool(args.miopen)

    # TPGAN_DDP_BACKEND=gloo: rehearsal of the multi-rank launch on a box with fewer GPUs than ranks
    # (ranks share devices, collectives go over gloo); the real run uses nccl = RCCL, one GPU per rank
    rank, world, local = ddp.init_from_env(backend=os.environ.get("TPGAN_DDP_BACKEND"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    if local >= torch.cuda.device_count():
        if os.environ.get("TPGAN_DDP_BACKEND") != "gloo":
            raise SystemExit(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
        log(f"rank {rank}: REHEARSAL -- sharing cuda:{local % torch.cuda.device_count()} with another rank")
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    amp_dtype = torch.bfloat16 if args.dtype == "bf16" else None
    sync = ddp.GradSync()

    np.random.seed(1234 + rank)
    models = build(device, capturable=not args.no_graph)
    sync.broadcast_state(*models[:3])
    clips = [fluid_clip(args.batch, args.points, 8, 3, seed=1234 + rank * 1000 + s, device=device)
             for s in range(4)]
    global GRAPHED, EAGER_BODY
    EAGER_BODY = bool(args.eager_body)
    mode = "eager"
    if not args.no_graph:
        try:
            from tpgan_amd.gan_step_graph import GraphedFluidStep
            G, Ds, Dt, opts = models
            GRAPHED = GraphedFluidStep(G, Ds, Dt, opts, OPT, clips[0][0], clips[0][1], 1.0, amp_dtype, sync)
            mode = "hipgraph (%d graph%s per step)" % (len(GRAPHED._graphs[True]), "s" if len(GRAPHED._graphs[True]) > 1 else "")
        except Exception as e:   # noqa: BLE001 -- never lose the measurement to a capture problem
            GRAPHED = None
            log(f"hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
        if world > 1:
            # the eager step issues other collectives than the replayed one (one all-reduce per
            # network instead of one flat one): either every rank replays or none does
            ok = torch.tensor([1.0 if GRAPHED is not None else 0.0], device=device)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if GRAPHED is not None and float(ok.item()) == 0.0:
                GRAPHED = None
                log(f"rank {rank}: another rank could not capture; running eagerly like it")
    if EAGER_BODY and GRAPHED is not None:
        mode = "captured step body, launched eagerly"
    log(f"rank {rank}: step mode = {mode}")

    log(f"rank {rank}: models and clips resident, warming up")
    for w in range(args.warmup):
        run_steps(models, clips, 1, sync, amp_dtype, start=w)
        torch.cuda.synchronize()
        log(f"warm-up step {w + 1}/{args.warmup} done")
    ddp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run_steps(models, clips, args.steps, sync, amp_dtype, start=args.warmup)
    torch.cuda.synchronize()
    ddp.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    dt = float(dt.item())
    log(f"timed region: {args.steps} steps in {dt:.3f} s")

    line = {
        "metric": "GAN train-steps/sec (G+D) on 4096-pt x3-frame clips",
        "value": world * args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype if args.dtype == "fp32" else "bf16",
        "data": "synthetic",
        "config": {"workload": f"cfg2: {args.points}-pt x3-frame fluid clips, batch {args.batch} per GPU, "
                               "full G+D adversarial step (SRNet(3,128) + FluidTempoDis(3) + "
                               "FluidSpatialDis, Adam), all-keep mask regime, gate open, even iteration",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "value_definition": "batch-of-%d steps per second summed over ranks" % args.batch,
                   "precision": "bf16 autocast on 1x1 convs/linears; coordinates, neighbour search, "
                                "indices, Chamfer in fp32" if args.dtype == "bf16" else "fp32",
                   "parallelism": f"dp{world}", "step_mode": mode, "last_losses": last},
    }
    if rank == 0 and world == 1 and not args.no_extra:
        roof, table = roofline_leg(models, clips, min(args.steps, 5), sync, amp_dtype)
        line["roofline"] = roof
        line["kernels"] = table
        log("roofline leg done, timing the CPU baseline")
        line["cpu_baseline"] = cpu_baseline(args.cpu_sample_batch, args.batch, args.points)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
