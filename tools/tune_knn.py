"""Variant sweep for the kNN kernels (csrc/knn.hip); see tools/tune_rowbn.py.

    python tools/tune_knn.py build      # here
    python tools/tune_knn.py run [reps] # GPU box: times every variant, checks them bit-equal
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd", "csrc")
VDIR = os.path.join(CSRC, "variants")
VARIANTS = {"one_query_per_wave": {"TPG_KNN_NO_TILE": 1}, "q2": {"TPG_KNN_TILE_Q": 2}, "q4": {"TPG_KNN_TILE_Q": 4},
            "q8": {"TPG_KNN_TILE_Q": 8}, "q16": {"TPG_KNN_TILE_Q": 16}}
# (B, P, D, K): the generator's feature-space searches at cfg2 with the T frames stacked
SHAPES = [(24, 512, 64, 12), (24, 512, 64, 4), (24, 512, 32, 40), (24, 512, 32, 20), (24, 512, 32, 9), (8, 512, 64, 12)]


def build():
    os.makedirs(VDIR, exist_ok=True)
    for tag, defs in VARIANTS.items():
        out = os.path.join(VDIR, f"knn_{tag}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                               "-fPIC", "-shared", "-I", os.path.join(ROOT, "include")] +
                              [f"-D{k}={v}" for k, v in defs.items()] + [os.path.join(CSRC, "knn.hip"), "-o", out])
        print("built", out)


def run(reps):
    import torch
    P_, I, F = C.c_void_p, C.c_int, C.c_float
    dev = torch.device("cuda", 0)
    for (B, P, D, K) in SHAPES:
        x = [torch.randn(B, P, D, device=dev) for _ in range(4)]
        ref = None
        print(f"\n== B={B} P={P} D={D} K={K}")
        for tag in VARIANTS:
            lib = C.CDLL(os.path.join(VDIR, f"knn_{tag}.so"))
            lib.tpg_knn_f32.argtypes = [P_, P_, P_, P_, I, I, I, I, I, F, P_, P_, P_]
            dist = torch.empty(B, P, K, device=dev)
            idx = torch.empty(B, P, K, device=dev, dtype=torch.int64)

            def go(t, st):
                rc = lib.tpg_knn_f32(t.data_ptr(), t.data_ptr(), None, None, B, P, P, D, K, -1.0, dist.data_ptr(),
                                     idx.data_ptr(), st)
                assert rc == 0, rc
            go(x[0], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if ref is None:
                ref = (dist.clone(), idx.clone())
            same = torch.equal(dist, ref[0]) and torch.equal(idx, ref[1])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cs = torch.cuda.current_stream().cuda_stream
                for i in range(reps):
                    go(x[i % 4], cs)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record()
            torch.cuda.synchronize()
            print("%-20s %8.1f us   bit-equal to the first variant: %s" % (tag, e0.elapsed_time(e1) * 1e3 / reps, same))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
