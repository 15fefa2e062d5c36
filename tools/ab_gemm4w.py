"""One wave per SIMD against two: the prototype k_gemm4w (debug build; 256 threads, 128 x 128 per wave, AGPR accumulators, one barrier per K tile,
bias epilogue) against the production persistent 256^2 kernel with the SAME bias-only bf16 epilogue and against hipBLASLt (F.linear), on the
four layer shapes; outputs of the two hand-written kernels compared bit for bit."""
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
    b = torch.randn(N, generator=g).to(dev); bh = b.to(torch.bfloat16)
    out4 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f4 = lambda: _lib.check(L.cpx_gemm4w(A.data_ptr(), W.data_ptr(), M, N, K, b.data_ptr(), out4.data_ptr(), N, st))
    f8 = lambda: ops.gemm(A, W, "bf16", b)
    fv = lambda: torch.nn.functional.linear(A, W, bh)
    t = {}
    for rnd in range(2):
        for k, f in ((("4w", f4), ("8w", f8), ("vendor", fv)) if rnd == 0 else (("vendor", fv), ("8w", f8), ("4w", f4))):
            t.setdefault(k, []).append(timeit(f))
    o8 = f8(); f4(); torch.cuda.synchronize()
    fl = 2.0 * M * N * K
    same = bool(torch.equal(o8.view(torch.int16), out4.view(torch.int16)))
    ref = A[:512].float() @ W.float().T + b
    err = float((out4[:512].float() - ref).norm() / ref.norm())
    print(f"{name:9s} {M}x{N}x{K}: one wave per SIMD {np.mean(t['4w']):7.1f} us ({fl / np.mean(t['4w']) / 1e6:6.0f} TF/s)   8-wave persistent {np.mean(t['8w']):7.1f} us"
          f"   hipBLASLt {np.mean(t['vendor']):7.1f} us   4w / 8w {np.mean(t['4w']) / np.mean(t['8w']):5.3f}   bitwise equal to the 8-wave kernel: {same}   rel. error vs f32 torch {err:.2e}")
