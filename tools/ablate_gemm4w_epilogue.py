"""What does the epilogue of the dominant kernel cost with the matrix pipe idle?  (Round-5 review item 1: "get epilogue(t) under main-loop(t + 1)".)

mlp.lin1 in its engine configuration (bf16, 16 384 x 4 096 x 1 024 per launch = one row part; folded LayerNorm + bias + erf-GELU) on
k_gemm4w, debug build, interleaved rounds in one process:
    full      the production kernel (output stores at agent scope, sc1)
    bias      the same main loop with the bias-only epilogue (accumulator read-out, one conversion per pair, lane-row swap, 16-byte stores)
    none      no epilogue at all (accumulators neither read nor stored; timing only)
    plain / sc0sc1 / nt   the production epilogue (agent-scope stores since round 6) with ordinary stores (round 5) / system scope / non-temporal
full - none = the epilogue's exposed time per launch = the MOST any scheme that hides it under the next tile's MFMAs can gain;
bias - none = the part of it that is read-out + stores; full - bias = LayerNorm fold + GELU arithmetic."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
nS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
def timeit(n=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
VARS = {"full": 0, "bias": 2049, "none": 2048, "plain": 4096, "sc0sc1": 8192, "nt": 512}
res = {k: [] for k in VARS}
for _ in range(5): f()
for rnd in range(9):
    order = list(VARS) if rnd % 2 == 0 else list(VARS)[::-1]
    for k in order:
        L.cpx_gemm4w_set_variant(VARS[k]); res[k].append(timeit())
L.cpx_gemm4w_set_variant(0)
med = {k: float(np.median(v)) for k, v in res.items()}
fl = 2.0 * M * N * K
tiles = (M // 256) * (N // 256)
print(f"mlp.lin1 row part {M} x {N} x {K} on k_gemm4w (debug build), 9 interleaved rounds of 40 launches, us per launch (median / min); {tiles} output tiles on 256 CUs = {tiles / 256:.0f} per workgroup")
for k in VARS:
    print(f"  {k:5s} {med[k]:7.1f} / {min(res[k]):7.1f}   {fl / med[k] / 1e6:7.1f} TFLOP/s = {fl / med[k] / 1e6 / 2500:.3f} of 2.5 PF")
print(f"  epilogue exposed: full - none = {med['full'] - med['none']:.1f} us = {100 * (med['full'] - med['none']) / med['full']:.1f} % of the launch "
      f"({(med['full'] - med['none']) / (tiles / 256):.2f} us per tile); read-out + stores (bias - none) {med['bias'] - med['none']:.1f} us, "
      f"LayerNorm fold + GELU (full - bias) {med['full'] - med['bias']:.1f} us")
print(f"  the review's target of 108 us per launch needs {med['full'] - 108:.1f} us = {100 * (med['full'] - 108) / max(med['full'] - med['none'], 1e-9):.0f} % of the exposed epilogue hidden")
