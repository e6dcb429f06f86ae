"""Variant sweep for the fused BatchNorm row kernels (csrc/rowbn.hip).

    python tools/tune_rowbn.py build            # here: hipcc each variant -> csrc/variants/*.so
    python tools/tune_rowbn.py run [reps]       # GPU box: time every variant on the step's shapes

Each variant is rowbn.hip alone compiled with a set of -D tunables; the tool calls the C-ABI
through ctypes (phase argument: one kernel per call) and times it with events on the launch
stream.  Buffers rotate over several copies so that the 256 MB MALL does not serve the data.
"""
import ctypes as C
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd", "csrc")
VDIR = os.path.join(CSRC, "variants")

VARIANTS = {
    "base": {},
    "mb128": {"TPG_BN_MAX_BLOCKS": 128},
    "mb512": {"TPG_BN_MAX_BLOCKS": 512},
    "mb1024": {"TPG_BN_MAX_BLOCKS": 1024},
    "div1": {"TPG_BN_SEG_DIV": 1},
    "div2": {"TPG_BN_SEG_DIV": 2},
    "mb512_div2": {"TPG_BN_MAX_BLOCKS": 512, "TPG_BN_SEG_DIV": 2},
    "sr4": {"TPG_BN_STATS_ROWS": 4},
    "sr16": {"TPG_BN_STATS_ROWS": 16},
    "su4": {"TPG_BN_STATS_U": 4},
    "acap512": {"TPG_BN_APPLY_CAP": 512},
    "acap1024": {"TPG_BN_APPLY_CAP": 1024},
    "bu4": {"TPG_BN_BWD_U": 4},
    "gcap1k": {"TPG_BN_GROUP_CAP": 1024},
    "gcap256": {"TPG_BN_GROUP_CAP": 256},
}
if os.environ.get("TUNE_ROWBN_ONLY"):      # e.g. TUNE_ROWBN_ONLY=base,round1k
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["TUNE_ROWBN_ONLY"].split(",")}
# (P rows of ALL segments, K, C, nseg): the shared-MLP tails of the discriminators at cfg2 (B=8) as the
# step launches them -- T frames x {fake, real} batches as segments
SHAPES = [(262144, 0, 64, 1), (524288, 0, 64, 2), (786432, 0, 64, 3), (1572864, 0, 64, 6),
          (524288, 32, 128, 2), (1572864, 32, 128, 6), (393216, 0, 128, 6), (393216, 32, 256, 6),
          (786432, 32, 128, 3), (1572864, 32, 256, 6), (245760, 20, 128, 1), (147456, 12, 256, 1)]


def build():
    os.makedirs(VDIR, exist_ok=True)
    for tag, defs in VARIANTS.items():
        out = os.path.join(VDIR, f"rowbn_{tag}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC",
               "-shared", "-I", os.path.join(ROOT, "include")] + [f"-D{k}={v}" for k, v in defs.items()] + \
              [os.path.join(CSRC, "rowbn.hip"), "-o", out]
        subprocess.check_call(cmd)
        print("built", out)


def run(reps):
    import torch
    P_, I, L, F = C.c_void_p, C.c_int, C.c_longlong, C.c_float
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    NSETS = 4
    results = {}
    for (P, K, Cc, nseg) in SHAPES:
        rows = P // K if K else P
        sets = []
        for _ in range(NSETS):
            x = torch.randn(P, Cc, device=dev).bfloat16()
            sets.append(dict(x=x, gy=torch.randn(rows, Cc, device=dev).bfloat16(), dx=torch.empty_like(x),
                             y=torch.empty(rows, Cc, device=dev, dtype=torch.bfloat16),
                             arg=torch.zeros(rows, Cc, device=dev, dtype=torch.uint8)))
        gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        mean, rstd = torch.empty(nseg * Cc, device=dev), torch.empty(nseg * Cc, device=dev)
        dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        eb = 2
        bytes_ = {"fwd_stats": P * Cc * eb, "fwd_apply": P * Cc * eb + rows * Cc * (eb + (1 if K else 0)),
                  "bwd_reduce": 2 * rows * Cc * eb if K else 2 * P * Cc * eb,
                  "bwd_apply": 2 * P * Cc * eb + rows * Cc * (eb + (1 if K else 0)) * (1 if K else 0)
                  + (P * Cc * eb if not K else 0)}
        for tag in VARIANTS:
            lib = C.CDLL(os.path.join(VDIR, f"rowbn_{tag}.so"))
            lib.tpg_rowbn_workspace_bytes.restype = C.c_size_t
            lib.tpg_rowbn_workspace_bytes.argtypes = [I, I]
            ws = torch.zeros(lib.tpg_rowbn_workspace_bytes(Cc, nseg), dtype=torch.uint8, device=dev)
            lib.tpg_rowbn_fwd.argtypes = [P_, I, L, I, I, F, F, I, P_, P_, P_, P_, P_, P_, F, P_, P_, P_, I, P_, P_, I, I, P_]
            lib.tpg_rowbn_bwd.argtypes = [P_, I, P_, I, P_, P_, I, L, I, I, I, P_, P_, P_, P_, F, P_, P_, P_, P_, I, I, P_]

            def fwd(s, phase, st=st):
                rc = lib.tpg_rowbn_fwd(s["x"].data_ptr(), 1, P, K, Cc, 1e-5, 0.1, 1, None, None, None, None,
                                       gamma.data_ptr(), beta.data_ptr(), 0.2, mean.data_ptr(), rstd.data_ptr(),
                                       s["y"].data_ptr(), 1, s["arg"].data_ptr() if K else None, ws.data_ptr(),
                                       nseg, phase, st)
                assert rc == 0, rc

            def bwd(s, phase, st=st):
                rc = lib.tpg_rowbn_bwd(s["gy"].data_ptr(), 1, s["x"].data_ptr(), 1, s["arg"].data_ptr() if K else None,
                                       s["y"].data_ptr() if K else None, 1, P, K, Cc, 1, mean.data_ptr(),
                                       rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0.2, dg.data_ptr(),
                                       db.data_ptr(), s["dx"].data_ptr(), ws.data_ptr(), nseg, phase, st)
                assert rc == 0, rc

            for s in sets:      # valid statistics / arg-max / y everywhere
                fwd(s, 0)
            for name, fn, phase in (("fwd_stats", fwd, 1), ("fwd_apply", fwd, 2), ("bwd_reduce", bwd, 1),
                                    ("bwd_apply", bwd, 2)):
                for s in sets:
                    fn(s, phase)
                torch.cuda.synchronize()
                # timed as a replayed hipGraph of `reps` launches, like the step itself (launch gaps
                # of the eager path would swamp the small shapes)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    cs = torch.cuda.current_stream().cuda_stream
                    for i in range(reps):
                        fn(sets[i % NSETS], phase, cs)
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                results[(P, K, Cc, nseg, name, tag)] = (us, bytes_[name] / us / 1e3)
        del sets
        torch.cuda.empty_cache()
    for (P, K, Cc, nseg) in SHAPES:
        print(f"\n== P={P} K={K} C={Cc} nseg={nseg} (bf16)   us per launch [GB/s; stats/reduce launches include their finalize]")
        print("%-16s" % "variant" + "".join("%22s" % n for n in ("fwd_stats", "fwd_apply", "bwd_reduce", "bwd_apply")))
        for tag in VARIANTS:
            row = "%-16s" % tag
            for n in ("fwd_stats", "fwd_apply", "bwd_reduce", "bwd_apply"):
                us, gbps = results[(P, K, Cc, nseg, n, tag)]
                row += "%12.1f [%6.0f]" % (us, gbps)
            print(row)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
