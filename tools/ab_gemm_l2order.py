"""A/B of the super-tile order of the 256^2 GEMMs: mode 1 (an XCD sweeps N for a fixed band of 8 row tiles) against
mode 2 (an XCD keeps one 4-tile W panel and streams the activation rows past it), on the per-layer shapes with the
production epilogues: interleaved rounds in one process + bitwise comparison."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M = 32768
g = torch.Generator().manual_seed(0)
shapes = [("qkv", 3072, 1024, "qkv", True), ("proj", 1024, 1024, "resid", False), ("fc1", 4096, 1024, "gelu", True), ("fc2", 1024, 4096, "resid", False)]
st = torch.cuda.current_stream().cuda_stream
for name, N, K, epi, ln in shapes:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    stats = ops.row_stats(A) if ln else None
    cs = W.float().sum(1).contiguous() if ln else None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    vt = torch.empty((M // 1024, 16, 64, 1024), dtype=torch.bfloat16, device=dev)
    sto = torch.zeros((M, 4, 2), dtype=torch.float32, device=dev)
    aux = vt if epi == "qkv" else (res if epi == "resid" else None)
    def run():
        _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI[epi], b.data_ptr(), aux.data_ptr() if aux is not None else None,
                                 out.data_ptr(), N, stats.data_ptr() if ln else None, cs.data_ptr() if ln else None,
                                 sto.data_ptr() if epi == "resid" else None, st))
    outs, t = {}, {2: [], 1: []}
    for p in (1, 2):
        L.cpx_gemm_set_l2_block(p)
        out.zero_(); vt.zero_(); sto.zero_()
        run(); torch.cuda.synchronize()
        outs[p] = (out.clone(), vt.clone(), sto.clone())
    same = all(torch.equal(a, b_) for a, b_ in zip(outs[2], outs[1]))
    for rnd in range(6):
        for p in (1, 2):
            L.cpx_gemm_set_l2_block(p)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            t[p].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * N * K
    for p in (1, 2):
        m = float(np.median(t[p]))
        print(f"{name:5s} N={N:5d} K={K:5d} {'N-sweep (1)' if p == 1 else 'M-sweep (2)'}: median {m:7.1f} us  min {min(t[p]):7.1f}  {fl / m / 1e6:7.1f} TFLOP/s   bitwise equal: {same}", flush=True)
L.cpx_gemm_set_l2_block(1)
