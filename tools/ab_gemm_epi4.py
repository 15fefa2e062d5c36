"""A/B of the quarter-tile epilogue of the persistent 256^2 GEMM (cpx_gemm_set_epi4: the conversion of quarter q runs beside the
LDS reads and stores of quarter q - 1, two 34 KB staging buffers) against the two-half epilogue: bitwise comparison,
repeatability under concurrent load, interleaved timing, on the per-layer shapes it covers (no residual epilogue yet)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
M = 32768
shapes = [("fc1", 4096, 1024, "gelu", True), ("qkv", 3072, 1024, "qkv", True), ("bf16 + bias", 2048, 2048, "bf16", False), ("relu K=4096", 1024, 4096, "relu", False)]
st = torch.cuda.current_stream().cuda_stream
for name, N, K, epi, ln in shapes:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    stats = ops.row_stats(A) if ln else None
    cs = W.float().sum(1).contiguous() if ln else None
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    vt = torch.empty((M // 1024, 16, 64, 1024), dtype=torch.bfloat16, device=dev)
    def run():
        _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI[epi], b.data_ptr(), vt.data_ptr() if epi == "qkv" else None,
                                 out.data_ptr(), N, stats.data_ptr() if ln else None, cs.data_ptr() if ln else None, None, st))
    outs, t = {}, {0: [], 1: []}
    for v in (0, 1):
        L.cpx_gemm_set_epi4(v)
        out.zero_(); vt.zero_(); run(); torch.cuda.synchronize()
        outs[v] = (out.clone(), vt.clone())
    same = all(torch.equal(a_, b_) for a_, b_ in zip(outs[0], outs[1]))
    L.cpx_gemm_set_epi4(1)
    noise = torch.empty((8192, 8192), device=dev); side = torch.cuda.Stream(dev); stable = True
    for i in range(12):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                noise.normal_()
        out.zero_(); run(); torch.cuda.synchronize()
        stable &= torch.equal(out, outs[1][0])
    side.synchronize()
    for rnd in range(6):
        for v in (0, 1):
            L.cpx_gemm_set_epi4(v)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / 20 * 1e3)
    for v in (0, 1):
        print(f"{name:12s} N={N:5d} K={K:5d} {'quarter epilogue' if v else 'two halves      '}: median {np.median(t[v]):7.1f} us  min {min(t[v]):7.1f}   "
              f"bitwise equal: {same}   repeatable: {stable}", flush=True)
L.cpx_gemm_set_epi4(0)
