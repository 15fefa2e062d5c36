"""Timing-only ablations of the 256^2 GEMM epilogue (interleaved, one process)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M = 32768
g = torch.Generator().manual_seed(0)
for name, N, K, epi in (("fc1", 4096, 1024, "gelu"), ("qkv-like", 3072, 1024, "bf16"), ("fc2-like", 1024, 4096, "bf16")):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = {}
    for rnd in range(3):
        for dbg, label in ((0, "full"), (2, "no GELU math"), (1, "no global stores"), (3, "neither"), (4, "no epilogue"), (8, "direct 16-B stores, no LDS staging")):
            L.cpx_gemm_set_dbg(dbg)
            for _ in range(3): ops.gemm(A, W, epi, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm(A, W, epi, b)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    L.cpx_gemm_set_dbg(0)
    print(name, {k: round(min(v), 1) for k, v in res.items()}, "ideal@1331TF", round(2.0 * M * N * K / 1331e12 * 1e6, 1), flush=True)
