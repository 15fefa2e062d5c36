"""A/B microbenchmark of the GEMM kernels on the four per-layer shapes (interleaved rounds)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops

dev = torch.device("cuda:0")
L = _lib.lib()
M = 32768
shapes = [("qkv", 3072, 1024, "bf16"), ("proj", 1024, 1024, "resid"), ("fc1", 4096, 1024, "gelu"), ("fc2", 1024, 4096, "resid")]
g = torch.Generator().manual_seed(0)
for name, N, K, epi in shapes:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev) if epi == "resid" else None
    res_t = {}
    for rnd in range(3):
        for big in (1, 0):
            L.cpx_gemm_set_big(big)
            for _ in range(3):
                ops.gemm(A, W, epi, b, res)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(A, W, epi, b, res)
            e1.record(); torch.cuda.synchronize()
            res_t.setdefault(big, []).append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    for big in (1, 0):
        t = min(res_t[big])
        print(f"{name:5s} N={N:5d} K={K:5d} {'256^2' if big else '128^2'}: {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s", flush=True)
L.cpx_gemm_set_big(1)
