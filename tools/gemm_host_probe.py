"""Host-side cost of F.linear per call for fp32 / bf16 under the two BLAS back ends (GPU box)."""
import sys
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.enabled = False
dev = "cuda"
for lib in ("cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:  # noqa: BLE001
        print(lib, "unavailable", e)
        continue
    for dt in (torch.float32, torch.bfloat16):
        for M, K, N in ((131072, 128, 128), (262144, 64, 128), (65536, 256, 128), (4096, 131, 128)):
            x = torch.randn(M, K, device=dev, dtype=dt)
            w = torch.randn(N, K, device=dev, dtype=dt)
            for _ in range(3):
                F.linear(x, w)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                F.linear(x, w)
            host = (time.perf_counter() - t0) / 50
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t0) / 50
            print(f"{lib:9s} {str(dt):15s} M{M} K{K} N{N}: host {host*1e6:8.1f} us/call, wall {tot*1e6:8.1f} us/call, "
                  f"{2*M*K*N/tot/1e12:6.1f} TF/s", flush=True)
    # autocast path
    x = torch.randn(131072, 128, device=dev)
    w = torch.randn(128, 128, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            F.linear(x, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            F.linear(x, w)
        host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
    print(f"{lib:9s} autocast bf16 fp32-in: host {host*1e6:8.1f} us/call", flush=True)
