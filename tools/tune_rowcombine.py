"""Variant sweep for the row-gather kernels (csrc/rowgather.hip); see tools/tune_rowbn.py.

    python tools/tune_rowcombine.py build      # here
    python tools/tune_rowcombine.py run [reps] # GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "temporal-pointcloud-upsampling-gan_amd", "csrc")
VDIR = os.path.join(CSRC, "variants")
VARIANTS = {
    "base": {},
    "nt": {"TPG_RC_NT_STORE": 1},
    "nt_u2": {"TPG_RC_NT_STORE": 1, "TPG_RC_FWD_U": 2},
    "u1": {"TPG_RC_FWD_U": 1, "TPG_RC_FWD_CAP": 16384},
    "u2": {"TPG_RC_FWD_U": 2},
    "cap4k": {"TPG_RC_FWD_CAP": 4096},
}
# (B, N, S, K, C, mode, dtype_in, dtype_out): 0 = f32, 1 = bf16; modes 0 gather, 1 sub, 2 edge
SHAPES = [(8, 4096, 1024, 32, 64, 1, 0, 1), (24, 4096, 1024, 32, 64, 1, 0, 1), (8, 1024, 256, 32, 128, 1, 0, 1),
          (8, 256, 256, 32, 128, 1, 0, 1), (8, 512, 512, 20, 32, 2, 0, 1), (16, 512, 512, 20, 32, 2, 0, 1),
          (8, 512, 512, 9, 32, 0, 0, 0), (16, 512, 512, 9, 32, 0, 0, 0)]


def build():
    os.makedirs(VDIR, exist_ok=True)
    for tag, defs in VARIANTS.items():
        out = os.path.join(VDIR, f"rowgather_{tag}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC",
               "-shared", "-I", os.path.join(ROOT, "include")] + [f"-D{k}={v}" for k, v in defs.items()] + \
              [os.path.join(CSRC, "rowgather.hip"), "-o", out]
        subprocess.check_call(cmd)
        print("built", out)


def run(reps):
    import torch
    P_, I, F = C.c_void_p, C.c_int, C.c_float
    dev = torch.device("cuda", 0)
    tdt = {0: torch.float32, 1: torch.bfloat16}
    NSETS = 4
    res = {}
    for shp in SHAPES:
        B, N, S, K, Cc, mode, din, dout = shp
        sets = []
        for i in range(NSETS):
            g = torch.Generator(device=dev).manual_seed(i)
            # ball-query-like lists: a few true neighbours, the rest repeats of the first hit
            idx = torch.randint(0, N, (B, S, K), device=dev, dtype=torch.int32, generator=g)
            nhit = torch.randint(1, K + 1, (B, S, 1), device=dev, generator=g)
            idx = torch.where(torch.arange(K, device=dev).view(1, 1, K) < nhit, idx, idx[:, :, :1]).contiguous()
            sets.append(dict(U=torch.randn(B, N, Cc, device=dev).to(tdt[din]),
                             Q=torch.randn(B, S, Cc, device=dev).to(tdt[din]), idx=idx,
                             out=torch.empty(B, S, K, Cc, device=dev, dtype=tdt[dout]),
                             gout=torch.randn(B, S, K, Cc, device=dev).to(tdt[dout]),
                             gU=torch.empty(B, N, Cc, device=dev, dtype=tdt[din]),
                             gQ=torch.empty(B, S, Cc, device=dev, dtype=tdt[din]),
                             offs=torch.empty(B, N + 1, device=dev, dtype=torch.int32),
                             lst=torch.empty(B, S * K, device=dev, dtype=torch.int32)))
        eo, ei = sets[0]["out"].element_size(), sets[0]["U"].element_size()
        rows = B * S * K
        bytes_ = {"fwd": rows * Cc * (eo + ei) + 4 * rows + (B * S * Cc * ei if mode else 0) + (rows * Cc * ei if mode == 2 else 0),
                  "bwd": rows * Cc * eo * (2 if mode == 1 else 1) + 8 * rows + B * (N + (S if mode else 0)) * Cc * ei
                  + (2 * rows * Cc * ei if mode == 2 else 0)}
        for tag in VARIANTS:
            lib = C.CDLL(os.path.join(VDIR, f"rowgather_{tag}.so"))
            lib.tpg_rowcombine_fwd.argtypes = [P_, P_, P_, I, I, I, I, I, I, I, I, F, P_, P_]
            lib.tpg_invert_index.argtypes = [P_, I, I, I, P_, P_, P_, P_]
            lib.tpg_rowcombine_bwd.argtypes = [P_, P_, P_, P_, P_, I, I, I, I, I, I, I, I, F, P_, P_, P_]

            def fwd(s, st):
                rc = lib.tpg_rowcombine_fwd(s["U"].data_ptr(), s["Q"].data_ptr() if mode else None, s["idx"].data_ptr(),
                                            mode, din, dout, B, N, S, K, Cc, 0.2, s["out"].data_ptr(), st)
                assert rc == 0, rc

            def bwd(s, st):
                rc = lib.tpg_rowcombine_bwd(s["gout"].data_ptr(), s["idx"].data_ptr(), s["offs"].data_ptr(),
                                            s["lst"].data_ptr(), s["Q"].data_ptr() if mode == 2 else None, mode, din,
                                            dout, B, N, S, K, Cc, 0.2, s["gU"].data_ptr(),
                                            s["gQ"].data_ptr() if mode else None, st)
                assert rc == 0, rc

            st0 = torch.cuda.current_stream().cuda_stream
            for s in sets:
                tmp = torch.empty(B, S * K, device=dev, dtype=torch.int32)
                assert lib.tpg_invert_index(s["idx"].data_ptr(), B, N, S * K, s["offs"].data_ptr(), s["lst"].data_ptr(),
                                            tmp.data_ptr(), st0) == 0
            for name, fn in (("fwd", fwd), ("bwd", bwd)):
                for s in sets:
                    fn(s, st0)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    cs = torch.cuda.current_stream().cuda_stream
                    for i in range(reps):
                        fn(sets[i % NSETS], cs)
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                res[(shp, name, tag)] = (us, bytes_[name] / us / 1e3)
        del sets
        torch.cuda.empty_cache()
    for shp in SHAPES:
        print(f"\n== B,N,S,K,C,mode,din,dout = {shp}   us per call [GB/s]  (bwd of mode 1 = 2 launches)")
        for tag in VARIANTS:
            print("%-10s" % tag + "".join("%8s %9.1f [%6.0f]" % (n, *res[(shp, n, tag)]) for n in ("fwd", "bwd")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
