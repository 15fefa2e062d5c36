"""mlp.lin1 (LayerNorm-less ops.gemm form, GELU epilogue) and a bias-only GEMM of the same shape over the number of token rows M:
does the time per row change when the 2 M-byte-per-row output stops fitting the 256 MB memory-side cache?  (debug build for the
store-less ablation; product kernels otherwise)"""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
N, K = 4096, 1024
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
print("mlp.lin1 shape, GELU epilogue: us per launch (us per 1024 rows) -- full kernel | without the global stores | main loop only")
for M in (8192, 16384, 24576, 32768, 49152, 65536):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    row = []
    for dbg in (32, 33, 4):
        L.cpx_gemm_set_dbg(dbg)
        ts = []
        for rnd in range(3):
            for _ in range(3): ops.gemm(A, W, "gelu", b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm(A, W, "gelu", b)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        row.append(float(np.median(ts)))
    L.cpx_gemm_set_dbg(0)
    print(f"  M {M:6d} (output {M * N * 2 / 1e6:5.0f} MB, {M // 256 * 16 / 256:.1f} tiles per CU): " + " | ".join(f"{t:7.1f} ({t / (M / 1024):5.2f})" for t in row))
