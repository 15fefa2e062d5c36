"""Race screen for the 256^2 GEMM: many launches, bitwise comparison against the first result and
against the 128^2 kernel, with a second stream hammering HBM meanwhile."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
junk = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)
bad = 0
for (M, N, K, epi) in ((32768, 3072, 1024, "bf16"), (32768, 1024, 1024, "resid"), (32768, 4096, 1024, "gelu"),
                       (32768, 1024, 4096, "resid"), (8192, 2048, 128, "bf16"), (16384, 1024, 256, "relu")):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev) if epi == "resid" else None
    L.cpx_gemm_set_big(0); ref128 = ops.gemm(A, W, epi, b, res); L.cpx_gemm_set_big(1)
    first = ops.gemm(A, W, epi, b, res)
    same128 = torch.equal(first, ref128)
    nbad = 0
    for it in range(150):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)                      # uneven background load on another stream
        out = ops.gemm(A, W, epi, b, res)
        if not torch.equal(out, first):
            nbad += 1
    torch.cuda.synchronize()
    bad += nbad
    print(f"{M}x{N}x{K} {epi}: mismatching repeats {nbad}/150, bitwise equal to the 128^2 kernel: {same128}", flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean")
