"""A/B of the qkv projection: one workgroup per tile (k_gemm256) against the persistent kernel with the balanced
q|k / V^T tile list (cpx_gemm_set_persistent_qkv), folded LayerNorm + V^T epilogue as in production; interleaved
rounds in one process + bitwise comparison, for 32, 96, 144 (a half round at the end) and 16 sub-tiles."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
N, K = 3072, 1024
for M in (32768, 98304, 147456, 16384):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    stats = ops.row_stats(A)
    cs = W.float().sum(1).contiguous()
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    vt = torch.empty((M // 1024, 16, 64, 1024), dtype=torch.bfloat16, device=dev)
    def run():
        _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["qkv"], b.data_ptr(), vt.data_ptr(),
                                 out.data_ptr(), N, stats.data_ptr(), cs.data_ptr(), None, st))
    outs, t = {}, {0: [], 1: []}
    for p in (1, 0):
        L.cpx_gemm_set_persistent_qkv(p)
        out.zero_(); vt.zero_()
        run(); torch.cuda.synchronize()
        outs[p] = (out.clone(), vt.clone())
    same = all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
    for rnd in range(6):
        for p in (1, 0):
            L.cpx_gemm_set_persistent_qkv(p)
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            t[p].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * N * K
    for p in (0, 1):
        m = float(np.median(t[p]))
        print(f"qkv M={M:6d} {'persistent, balanced' if p else 'one workgroup per tile'}: median {m:7.1f} us  min {min(t[p]):7.1f}  {fl / m / 1e6:7.1f} TFLOP/s   bitwise equal: {same}", flush=True)
L.cpx_gemm_set_persistent_qkv(1)
