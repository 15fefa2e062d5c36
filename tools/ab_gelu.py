"""mlp.lin1 (persistent 256^2 GEMM, folded LayerNorm + GELU) with the production GELU (2 ^ P5(|x|) form, 7 VALU + 1
transcendental slots, tools/fit_gelu.py) against the Abramowitz-Stegun 7.1.26 form used until round 2 (11 + 2 slots) and
against no GELU at all: timing only, all three in the DBG instantiation, interleaved."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M, N, K = 32768, 4096, 1024
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
stats = ops.row_stats(A); cs = W.float().sum(1).contiguous()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["gelu"], b.data_ptr(), None, out.data_ptr(), N,
                             stats.data_ptr(), cs.data_ptr(), None, st))
variants = {"2^P5 form (production)": 32, "A-S 7.1.26 (round 2)": 32 | 16, "no GELU": 32 | 2}
t = {k: [] for k in variants}
for rnd in range(6):
    for k, d in variants.items():
        L.cpx_gemm_set_dbg(d)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        t[k].append(e0.elapsed_time(e1) / 20 * 1e3)
L.cpx_gemm_set_dbg(0)
for k, v in t.items():
    print(f"{k:26s} median {np.median(v):7.1f} us  min {min(v):7.1f}")
