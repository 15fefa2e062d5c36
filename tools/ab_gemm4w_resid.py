"""attn.proj and mlp.lin2 in their in-engine configuration (bf16, bias + residual in place + row statistics, M = 32 768): the one-wave-per-SIMD
kernel (csrc/cpx_gemm4w.hip) against the 8-wave persistent kernel (cpx_gemm_set_4w(0), debug build), interleaved rounds in one process,
outputs and statistics compared bit for bit."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M, N = 32768, 1024
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for name, K in (("attn.proj", 1024), ("mlp.lin2", 4096)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    x0 = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    x = x0.clone(); stats = torch.zeros((M, 4, 2), dtype=torch.float32, device=dev)
    def f():
        _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["resid"], b.data_ptr(), x.data_ptr(), x.data_ptr(), N, None, None, stats.data_ptr(), st))
    def timeit(n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    outs = {}
    for v in (3, 0):
        L.cpx_gemm_set_4w(v); x.copy_(x0); f(); torch.cuda.synchronize(); outs[v] = (x.clone(), stats.clone())
    res = {0: [], 3: []}
    for _ in range(3): f()
    for rnd in range(8):
        for v in ((3, 0) if rnd % 2 == 0 else (0, 3)):
            L.cpx_gemm_set_4w(v)
            res[v].append(timeit())
    L.cpx_gemm_set_4w(3)
    fl = 2.0 * M * N * K
    m4, m8 = float(np.median(res[3])), float(np.median(res[0]))
    same = bool(torch.equal(outs[0][0], outs[3][0])) and bool(torch.equal(outs[0][1], outs[3][1]))
    print(f"{name:9s} {M}x{N}x{K} (resid + stats): one wave per SIMD {m4:7.1f} us ({fl / m4 / 1e6 / 2500:.3f} of 2.5 PF)   8-wave persistent {m8:7.1f} us ({fl / m8 / 1e6 / 2500:.3f})   "
          f"4w / 8w {m4 / m8:.3f}   outputs and statistics bitwise equal: {same}", flush=True)
