"""Timing-only ablations of the persistent 256^2 GEMM MAIN LOOP (debug build, DBG instantiation; results are garbage): what the
loop costs without its LDS-DMA requests, without its fragment reads, without its barriers -- on the mlp.lin1 and mlp.lin2 shapes."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M = 32768
g = torch.Generator().manual_seed(0)
CASES = ((4, "main loop only (no epilogue)"), (4 | 64, "  ... without LDS-DMA requests"), (4 | 128, "  ... without fragment reads"),
         (4 | 64 | 128, "  ... without both (MFMAs + barriers)"), (4 | 256, "  ... without barriers"), (4 | 512, "  ... every tile reads activation rows 0..255 (cache-resident)"), (4 | 64 | 128 | 256, "  ... MFMAs alone"), (32, "full kernel"), (32 | 512, "full kernel, activation rows 0..255 for every tile"))
for name, N, K, epi in (("mlp.lin1", 4096, 1024, "gelu"), ("mlp.lin2-like", 1024, 4096, "bf16")):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = {}
    for rnd in range(4):
        for dbg, label in CASES:
            L.cpx_gemm_set_dbg(dbg)
            for _ in range(3): ops.gemm(A, W, epi, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm(A, W, epi, b)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(label, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    L.cpx_gemm_set_dbg(0)
    fl = 2.0 * M * N * K
    print(f"{name} ({M} x {N} x {K}); MFMA time at 2.5 PF: {fl / 2.5e15 * 1e6:.1f} us")
    for dbg, label in CASES:
        t = float(np.median(res[label]))
        print(f"   {label:44s} {t:7.1f} us   {fl / t / 1e6:7.1f} TFLOP/s")
