"""mlp.lin1 in its in-engine configuration (bf16, folded LayerNorm + bias + erf-GELU, 32 768 x 4 096 x 1 024): the one-wave-per-SIMD kernel
(csrc/cpx_gemm4w.hip, production since round 5) against the 8-wave persistent kernel it replaced (cpx_gemm_set_4w(0), debug build), interleaved
rounds in one process, outputs compared bit for bit.  Optional argument: number of sub-tiles (default 32)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
nS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M, N, K = nS * 1024, 4096, 1024
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
stats = ops.row_stats(A); colsum = W.float().sum(1).contiguous()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def f():
    _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["gelu"], b.data_ptr(), None, out.data_ptr(), N, stats.data_ptr(), colsum.data_ptr(), None, st))
def timeit(n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {0: [], 1: []}
outs = {}
for v in (1, 0):
    L.cpx_gemm_set_4w(v); f(); torch.cuda.synchronize(); outs[v] = out.clone()
for _ in range(3): f()
for rnd in range(8):
    for v in ((1, 0) if rnd % 2 == 0 else (0, 1)):
        L.cpx_gemm_set_4w(v)
        res[v].append(timeit())
L.cpx_gemm_set_4w(1)
fl = 2.0 * M * N * K
m4, m8 = float(np.median(res[1])), float(np.median(res[0]))
print(f"mlp.lin1 {M}x{N}x{K} (LN + GELU), 8 interleaved rounds of 20 launches (us per launch: median / min)")
print(f"  one wave per SIMD (k_gemm4w)   {m4:7.1f} / {min(res[1]):7.1f}   {fl / m4 / 1e6:7.1f} TFLOP/s = {fl / m4 / 1e6 / 2500:.3f} of 2.5 PF")
print(f"  8-wave persistent (k_gemm256p) {m8:7.1f} / {min(res[0]):7.1f}   {fl / m8 / 1e6:7.1f} TFLOP/s = {fl / m8 / 1e6 / 2500:.3f} of 2.5 PF")
print(f"  4w / 8w {m4 / m8:.3f}   outputs bitwise equal: {bool(torch.equal(outs[0], outs[1]))}")
