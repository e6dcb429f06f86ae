"""Headline benchmark: GAN train-steps/s (full G+D adversarial step) on synthetic
4096-point x 3-frame fluid clips, batch 8 per GPU (BASELINE.json configs[1]; weak scaling).

    python bench.py --gpus 1 --steps 200 --warmup 5
    python bench.py --config cfg4 | cfg5shard        # the other single-GPU workloads of BASELINE.json
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = `tempo_gan_step` semantics with generator AND both discriminators updated
(even iteration > 10, gate open, all-keep mask regime -- SURVEY.md section 8d).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N == 1 only):
  roofline      the hand-written kernel with the largest TOTAL time among ALL hand-written kernels,
                timed live with HIP events on its launch stream while the captured step body runs
                launch by launch; `roofline_streaming` is the same for the largest HBM-streaming
                kernel, `mfma` the matrix-core work (hand-written MFMA kernels and the remaining
                library GEMMs: flops / time against the 2.5 PFLOP/s dense bf16 peak);
  cpu_baseline  the same step function on the host cores through the oracle ops (kind
                "port": the reference has no CPU path for pointnet2_ops/FRNN), BASELINE.md section 3's
                protocol -- 2 warm-up steps + 5 timed steps, median -- on a bounded sample (2 clips per
                step instead of 8), scaled to the metric's unit.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import tpgan_amd  # noqa: E402
from tpgan_amd import configs, ddp, ops  # noqa: E402

T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBPS = 8000.0       # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16, same guide (AMD's 5 PF figure is 2:1 sparse)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 / 32x32x2: the fp32 vector rate (same guide)
CONFIG = "cfg2"


def build(device, seed=1, capturable=False):
    return configs.build_models(CONFIG, device, seed=seed, capturable=capturable)


GRAPHED = None   # GraphedFluidStep / GraphedActionStep when the hipGraph path is active
EAGER_BODY = False   # --eager-body: run the captured step body launch by launch (for PMC passes)
TRACE = bool(os.environ.get("TPGAN_BENCH_TRACE"))


def run_steps(models, clips, n, sync, amp_dtype, start=0):
    out = None
    for i in range(n):
        low, high = clips[(start + i) % len(clips)]
        if GRAPHED is not None and low[0].is_cuda and EAGER_BODY:
            out = GRAPHED.run_body_eagerly(low, high)
        elif GRAPHED is not None and low[0].is_cuda:
            out = GRAPHED(low, high, 12)
        else:
            out = configs.eager_step(CONFIG, models, (low, high), 12, sync=sync, amp_dtype=amp_dtype)
        if TRACE:
            log("step %d: %s" % (start + i, {k: round(v, 4) for k, v in out.items()}))
    return out


def cpu_baseline(sample_batch, per_gpu_batch, n_hi, warm=2, timed=5):
    """the same workload's step on the host cores (oracle ops + CPU PyTorch), 2 warm-up steps +
    5 timed steps, median; `sample_batch` clips per step (the batch that keeps the leg to about a minute), scaled to
    the batch-`per_gpu_batch` step."""
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
        clips = [configs.make_clip(CONFIG, batch=sample_batch, points=n_hi, seed=1234 + i) for i in range(2)]
        times = []
        for i in range(warm + timed):
            t0 = time.perf_counter()
            run_steps(models, clips, 1, None, None, start=i)
            dt = time.perf_counter() - t0
            if i >= warm:
                times.append(dt)
            log(f"cpu baseline step {i + 1}/{warm + timed}: {dt:.2f} s{'' if i >= warm else ' (warm-up)'}")
        med = float(np.median(times))
    finally:
        torch_backend.uninstall()
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    return {"value": (sample_batch / per_gpu_batch) / med, "unit": "steps/s", "cores": cores,
            "kind": "port", "cpu": model,
            "sample": f"{warm} warm-up + {timed} timed full G+D steps on {sample_batch} clips of {n_hi} pts x"
                      f"{configs.SPECS[CONFIG]['frames']} frames, median {med:.2f} s per step (min {min(times):.2f}, max "
                      f"{max(times):.2f}), scaled by {sample_batch}/{per_gpu_batch} to the batch-{per_gpu_batch} step; fp32; "
                      f"oracle C ops (OpenMP, {cores} threads) + CPU PyTorch convs"}


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
    library = [k for k in summ if k.startswith("gemm_")]          # hipBLASLt through torch, not hand-written
    hand = [k for k in summ if k not in library]
    # Streaming kernels are priced against the HBM roofline.  Furthest-point sampling and the
    # neighbour searches are chains of dependent rounds on cache-resident clouds (latency / issue
    # bound by construction): they get the same arithmetic -- algorithmic bytes / time -- and a note.
    streaming = [k for k in hand if k.startswith(("rowbn_", "rowcombine_", "group_", "mlp_"))]
    steps_f = float(steps)

    def roof(name, selection):
        d = summ[name]
        if name == "knn_mfma":
            # the matrix-core kNN filter streams a cache-resident cloud: priced against the matrix peak on its EXECUTED
            # flops (two sweeps, three split-bf16 products); what bounds it is the vector work of the selection
            return {"bound": "mfma", "kernel": name, "achieved": round(d["tflops"], 2), "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(d["tflops"] / MFMA_PEAK_TFLOPS, 6), "traffic": None,
                    "avg_launch_us": round(d["avg_us"], 2), "launches_per_step": d["launches"] / steps_f,
                    "ms_per_step": round(d["total_ms"] / steps_f, 4),
                    "executed_flops_per_launch": int(d["flops_per_launch"]), "selection": selection,
                    "note": "split-bf16 Gram filter + exact re-rank (DESIGN.md section 4d): ~115 vector instructions per 32 x 32 "
                            "pairs and sweep bound it, the 7 matrix instructions co-execute; launch time includes the fallback kernel"}
        return {"bound": "hbm", "kernel": name, "achieved": round(d["gbps"], 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(d["gbps"] / HBM_PEAK_GBPS, 6), "traffic": pmc_traffic(name),
                "traffic_source": "profiles/%s (rocprofv3 PMC passes of the cfg2 step body, average over the kernel's launches; "
                                  "not re-measured in this run; null for other configs)" % PMC_FILE,
                "avg_launch_us": round(d["avg_us"], 2), "launches_per_step": d["launches"] / steps_f,
                "ms_per_step": round(d["total_ms"] / steps_f, 4),
                "algorithmic_bytes_per_launch": int(d["bytes_per_launch"]), "selection": selection}
    dom = max(hand, key=lambda k: summ[k]["total_ms"])
    roofline = roof(dom, "largest total time among ALL hand-written kernels; timed per launch with HIP events "
                         "while the captured step body runs eagerly (launch by launch)")
    if dom == "fps":
        spec = configs.SPECS[CONFIG]
        roofline["note"] = ("furthest-point sampling: npoint-1 dependent rounds per launch on one workgroup per cloud, "
                            "bound by VALU issue + one barrier per round, not by bytes (DESIGN.md section 4: "
                            "0.57 us/round at 4096 points); hidden on the index-plan streams in graph mode")
        roofline["us_per_round"] = round(summ["fps"]["avg_us"] / max(1, spec["points"] // 4 - 1), 4)
    line = {"roofline": roofline}
    # north_star's own target: HBM utilisation of ball_query + group (here: the row gather that replaces
    # grouping_operation + first conv, csrc/rowgather.hip) -- algorithmic bytes of both over the time of both
    bq = [k for k in ("ball_query", "rowcombine_fwd") if k in summ]
    if len(bq) == 2:
        nbytes = sum(summ[k]["bytes_per_launch"] * summ[k]["launches"] for k in bq)
        us = sum(summ[k]["total_ms"] for k in bq) * 1e3
        line["ball_query_plus_group"] = {
            "bytes_per_step": int(nbytes / steps_f), "us_per_step": round(us / steps_f, 2),
            "achieved_GBps": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4),
            "parts": {k: {"launches_per_step": summ[k]["launches"] / steps_f, "avg_us": round(summ[k]["avg_us"], 2),
                          "GBps": round(summ[k]["gbps"], 1)} for k in bq},
            "note": "HIP events around each launch of the eagerly launched step body; target of BASELINE.json: 0.40"}
    if streaming:
        sdom = max(streaming, key=lambda k: summ[k]["total_ms"])
        line["roofline_streaming"] = roof(sdom, "largest total time among the HBM-streaming hand-written kernels")
    # matrix-core work: the hand-written MFMA kernels and what is left on the library
    mf = {}
    lib_ms = _library_gemm_ms(CONFIG)
    for grp, names in (("hand_written_mfma", [k for k in hand if k in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad", "knn_mfma")]),
                       ("hand_written_mfma_f32", [k for k in hand if k in ("small_tail_fwd", "small_tail_bwd")]),
                       ("library_gemm", library)):
        fl = sum(summ[k]["flops_per_launch"] * summ[k]["launches"] for k in names)
        ms = sum(summ[k]["total_ms"] for k in names)
        if ms > 0 and grp != "library_gemm":
            peak = MFMA_PEAK_TFLOPS if grp == "hand_written_mfma" else MFMA_F32_PEAK_TFLOPS
            mf[grp] = {"kernels": names, "tflop_per_step": round(fl / steps_f / 1e12, 4), "ms_per_step": round(ms / steps_f, 4),
                       "achieved_tflops": round(fl / 1e12 / (ms / 1e3), 2), "peak_tflops": peak,
                       "frac_of_peak": round(fl / 1e12 / (ms / 1e3) / peak, 5)}
        elif ms > 0:
            # HIP events around a torch GEMM call include the host's enqueue gap in this launch-by-launch leg
            # (20+ us of host time per call): the kernel time comes from the committed rocprofv3 summary
            mf[grp] = {"kernels": names, "tflop_per_step": round(fl / steps_f / 1e12, 4),
                       "launches_per_step": sum(summ[k]["launches"] for k in names) / steps_f,
                       "ms_per_step": lib_ms, "ms_source": "profiles/%s (Cijk_* rows / executions of %s; null when the "
                                                             "summaries were made from other sources)" % (STATS_FILE % CONFIG, META_FILE),
                       "achieved_tflops": round(fl / steps_f / 1e12 / (lib_ms / 1e3), 2) if lib_ms else None,
                       "frac_of_peak": round(fl / steps_f / 1e12 / (lib_ms / 1e3) / MFMA_PEAK_TFLOPS, 5)
                       if lib_ms else None}
    if mf:
        mf["peak_tflops"] = MFMA_PEAK_TFLOPS
        mf["note"] = ("bf16 MFMA, dense peak (the f32 matrix instructions of the 16-channel tails: the 157 TFLOP/s vector "
                      "rate); the MLP contractions run at 43-128 flop/B against a ridge of ~310 flop/B, i.e. they are "
                      "HBM-bound by design (see roofline_streaming); knn_mfma counts EXECUTED flops (two sweeps, three "
                      "split-bf16 products) and is bound by the vector work of its selection")
        line["mfma"] = mf
    table = {k: {"launches_per_step": v["launches"] / steps_f, "ms_per_step": round(v["total_ms"] / steps_f, 4),
                 "avg_us": round(v["avg_us"], 2), "GBps": round(v["gbps"], 1),
                 **({"TFLOPs": round(v["tflops"], 1)} if v["flops_per_launch"] else {})} for k, v in summ.items()}
    if "fps" in summ:
        table["fps"]["note"] = "npoint-1 dependent rounds per launch; hidden on a side stream in graph mode"
    for k in table:
        if k.startswith("gemm_"):
            # a torch GEMM call in the launch-by-launch body: the events bracket the host's enqueue work as well (plan
            # lookup, workspace, split-K views), the GPU idle in between -- NOT kernel time
            table[k]["note"] = ("HIP events around a torch / hipBLASLt call in the eager body include the host's enqueue gap; "
                                "kernel time of the library GEMMs: mfma.library_gemm (rocprofv3 summary of the replayed step)")
    return line, table


PMC_FILE = "r03_pmc_traffic.json"
STATS_FILE = "r03_final_%s_graph_bf16_kernel_stats.csv"      # % config
META_FILE = "r03_final_meta.json"       # written by tools/refresh_profiles.sh next to the summaries it describes


def source_fingerprint():
    """sha256 over the sources a kernel summary depends on (the library's csrc/, the package's Python, bench.py):
    tools/refresh_profiles.sh stores it next to the summaries, and figures read from them are reported only while it
    still matches (a code change must not leave stale profile numbers in the bench line)."""
    import glob
    import hashlib
    pkg = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.h*")) + glob.glob(os.path.join(pkg, "*.py"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "bench.py")])
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _profile_meta():
    """{"executions": step executions per profiled process, "source": fingerprint} of the committed summaries, or None
    when absent / made from other sources than the ones running now."""
    try:
        with open(os.path.join(ROOT, "profiles", META_FILE)) as fh:
            meta = json.load(fh)
        return meta if meta.get("source") == source_fingerprint() and meta.get("executions") else None
    except (OSError, ValueError):
        return None


def _library_gemm_ms(config):
    """hipBLASLt kernel time per step from the committed rocprofv3 summary of this config, or None when the file is
    not there or was made from another state of the sources."""
    import csv
    meta = _profile_meta()
    if meta is None:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", STATS_FILE % config)) as fh:
            tot = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(fh) if r["Name"].startswith("Cijk"))
        return round(tot / 1e6 / int(meta["executions"]), 4)
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), with the
    gfx950 FETCH_SIZE x2 correction for wide coalesced reads; None when no PMC file is present."""
    if CONFIG != "cfg2" or _profile_meta() is None:
        return None                      # the committed PMC passes are of the cfg2 step body, at a known source state
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    try:
        with open(path) as fh:
            rec = json.load(fh).get(kernel)
        return None if rec is None else rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)        # >= 2 s timed region at cfg2
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(configs.SPECS), default="cfg2",
                    help="BASELINE.json workload (cfg2 = the headline metric; cfg3 is cfg2 on 8 ranks)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the workload's, 8)")
    ap.add_argument("--points", type=int, default=None, help="high-res points per frame (default: the workload's)")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-extra", action="store_true", help="skip roofline and cpu_baseline legs")
    ap.add_argument("--cpu-sample-batch", type=int, default=2,
                    help="clips per CPU-baseline step (7 steps are run: 2 warm-up + 5 timed)")
    ap.add_argument("--no-graph", action="store_true",
                    help="run the step eagerly instead of replaying it from captured hipGraphs")
    ap.add_argument("--eager-body", action="store_true",
                    help="build the graphed step but run its body launch by launch (for rocprofv3 PMC "
                         "passes: per-kernel counters of the exact launches the graphs replay)")
    ap.add_argument("--miopen", action="store_true",
                    help="let PyTorch use MIOpen for conv/BN (first use JIT-compiles per shape: minutes)")
    args = ap.parse_args()
    global CONFIG
    CONFIG = args.config
    spec = configs.SPECS[CONFIG]
    args.batch = spec["batch"] if args.batch is None else args.batch
    args.points = spec["points"] if args.points is None else args.points
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
    clips = [configs.make_clip(CONFIG, batch=args.batch, points=args.points, seed=1234 + rank * 1000 + s, device=device)
             for s in range(4)]
    global GRAPHED, EAGER_BODY
    EAGER_BODY = bool(args.eager_body)
    mode = "eager"
    if not args.no_graph:
        try:
            GRAPHED = configs.graphed_step(CONFIG, models, clips[0], amp_dtype=amp_dtype, sync=sync)
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
    if world > 1 and GRAPHED is not None:
        GRAPHED.timing = []          # per step: [first graph, gradient all-reduce, second graph] in us (HIP events)
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
        "metric": "GAN train-steps/sec (G+D) on %d-pt x%d-frame clips" % (args.points, spec["frames"]),
        "value": world * args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype if args.dtype == "fp32" else "bf16",
        "data": "synthetic",
        "config": {"workload": spec["label"] + f"; batch {args.batch} per GPU",
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                   "value_definition": "batch-of-%d steps per second summed over ranks" % args.batch,
                   "precision": "bf16 autocast on 1x1 convs/linears; coordinates, neighbour search, "
                                "indices, Chamfer in fp32" if args.dtype == "bf16" else "fp32",
                   "parity": "asserted by tests/ (-m gpu), not restated here: kernels bit-exact vs oracle/tpgref.c, models vs "
                             "the reference goldens, replay bitwise == its body, bf16 vs fp32 conditioned "
                             "(tests/test_graph_gpu.py)",
                   "parallelism": f"dp{world}", "step_mode": mode, "last_losses": last},
    }
    if world > 1 and GRAPHED is not None and getattr(GRAPHED, "timing", None):
        # what one step of this rank spent where: explains the scaling number the driver computes
        t = torch.tensor(GRAPHED.timing, dtype=torch.float64, device=device).mean(0)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        names = ["graph_grads_us", "allreduce_us", "graph_apply_us"]
        line["multi_gpu_breakdown"] = {n: round(float(v), 1) for n, v in zip(names, t.tolist())}
        line["multi_gpu_breakdown"]["note"] = ("mean over the timed steps, max over ranks; HIP events on the step's stream: "
                                               "forward+backward graph, the ONE flat gradient all-reduce (RCCL), optimizer graph")
        log(f"rank {rank}: per-step breakdown {line['multi_gpu_breakdown']}")
    if rank == 0 and world == 1 and not args.no_extra:
        extra, table = roofline_leg(models, clips, min(args.steps, 5), sync, amp_dtype)
        line.update(extra)
        line["kernels"] = table
        log("roofline leg done, timing the CPU baseline")
        line["cpu_baseline"] = cpu_baseline(args.cpu_sample_batch, args.batch, args.points)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
