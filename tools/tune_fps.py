"""Workgroup shape sweep for furthest point sampling (csrc/fps.hip); see tools/tune_rowbn.py.

    python tools/tune_fps.py build | run
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd", "csrc")
VDIR = os.path.join(CSRC, "variants")
NOSLP = ["-fno-slp-vectorize"]
VARIANTS = {"default": {"_flags": NOSLP},
            "poll": {"_flags": NOSLP, "TPG_FPS_POLL_EXCHANGE": 1},
            "slp_poll": {"TPG_FPS_POLL_EXCHANGE": 1},
            "slp_barrier": {},          # the round-2 kernel (wrong picks on some hosts, see build.py)
            "w512": {"_flags": NOSLP, "TPG_FPS_4096_BLOCK": 512},
            "w1024": {"_flags": NOSLP, "TPG_FPS_4096_BLOCK": 1024, "TPG_FPS_2048_BLOCK": 512, "TPG_FPS_1024_BLOCK": 256},
            "narrow": {"_flags": NOSLP, "TPG_FPS_1024_BLOCK": 64, "TPG_FPS_2048_BLOCK": 128, "TPG_FPS_4096_BLOCK": 256, "TPG_FPS_8192_BLOCK": 512}}
SHAPES = [(24, 4096, 1024), (24, 1024, 256), (8, 1024, 512), (8, 2048, 512), (8, 8192, 1024), (8, 512, 128), (20, 16384, 1024)]


def build():
    os.makedirs(VDIR, exist_ok=True)
    for tag, defs in VARIANTS.items():
        out = os.path.join(VDIR, f"fps_{tag}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17",
                               "-fPIC", "-shared", "-I", os.path.join(ROOT, "include")] +
                              defs.get("_flags", []) + [f"-D{k}={v}" for k, v in defs.items() if k != "_flags"] +
                              [os.path.join(CSRC, "fps.hip"), "-o", out])
        print("built", out)


def run():
    import torch
    P_, I = C.c_void_p, C.c_int
    dev = torch.device("cuda", 0)
    for (B, N, m) in SHAPES:
        x = torch.rand(B, N, 3, device=dev) - 0.5
        ref = None
        print(f"\n== B={B} N={N} m={m}")
        for tag in VARIANTS:
            lib = C.CDLL(os.path.join(VDIR, f"fps_{tag}.so"))
            lib.tpg_fps_f32.argtypes = [P_, I, I, I, P_, P_, P_]
            idx = torch.empty(B, m, device=dev, dtype=torch.int32)
            st = torch.cuda.current_stream().cuda_stream
            assert lib.tpg_fps_f32(x.data_ptr(), B, N, m, None, idx.data_ptr(), st) == 0
            torch.cuda.synchronize()
            if ref is None:
                ref = idx.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.tpg_fps_f32(x.data_ptr(), B, N, m, None, idx.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            print("%-10s %8.1f us  (%.3f us/round)  same picks: %s" % (tag, us, us / (m - 1), torch.equal(idx, ref)))


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
