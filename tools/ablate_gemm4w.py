"""Timing-only ablations of the one-wave-per-SIMD GEMM's main loop (debug build, cpx_gemm4w_set_variant): 1 = no LDS-DMA requests inside the
loop, 2 = no fragment reads, 4 = no barriers, 8 = no MFMAs (results are garbage for every variant but 0), on the four layer shapes, beside the
8-wave persistent kernel with the same bias-only epilogue."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M = 32768
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,7,8".split(","))]
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
def timeit(fn, rounds=5, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return float(np.median(ts))
for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("mlp.lin1", 4096, 1024), ("mlp.lin2", 1024, 4096)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out4 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f4 = lambda: _lib.check(L.cpx_gemm4w(A.data_ptr(), W.data_ptr(), M, N, K, b.data_ptr(), out4.data_ptr(), N, st))
    f8 = lambda: ops.gemm(A, W, "bf16", b)
    acc = {}
    timeit(f8)
    for rnd in range(ROUNDS):                      # interleaved rounds, order reversed every other round
        order = [("8w", None)] + [(f"4w/{v}", v) for v in variants]
        for k, v in (order if rnd % 2 == 0 else order[::-1]):
            if v is None:
                acc.setdefault(k, []).append(timeit(f8))
            else:
                L.cpx_gemm4w_set_variant(v)
                acc.setdefault(k, []).append(timeit(f4))
    L.cpx_gemm4w_set_variant(0)
    res = {k: float(np.median(v)) for k, v in acc.items()}
    ideal = 2.0 * M * N * K / 2.5e15 * 1e6
    print(f"{name:9s} (MFMA time at 2.5 PF: {ideal:6.1f} us)  " + "  ".join(f"{k} {v:6.1f}" for k, v in res.items()), flush=True)
