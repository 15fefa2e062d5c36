"""Race screen of the persistent 256^2 kernel's debug-build variants: the direct-store epilogue and the balanced fragment-read schedule forced
on every epilogue (cpx_gemm_set_direct / cpx_gemm_set_balanced = 2), repeated launches under a concurrent memory-bound stream, every output
compared bit for bit with the production form's.   python tools/stress_gemm_variants.py [repeats [MxNxK]]"""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
bad_total = 0
SHAPES = ((8192, 2048, 1024), (16384, 2048, 256), (16384, 1024, 512), (32768, 1024, 4096))
if len(sys.argv) > 2:
    SHAPES = (tuple(int(x) for x in sys.argv[2].split("x")),)
for M, N, K in SHAPES:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    run = lambda: [ops.gemm(A, W, "bf16", bias), ops.gemm(A, W, "gelu", bias), ops.gemm(A, W, "resid", bias, res)]
    L.cpx_gemm_set_direct(0); L.cpx_gemm_set_balanced(0)
    ref = run()
    for direct, bal in ((2, 0), (0, 1), (2, 1), (1, 1)):
        L.cpx_gemm_set_direct(direct); L.cpx_gemm_set_balanced(bal)
        bad = [0, 0, 0]
        for r in range(reps):
            with torch.cuda.stream(side):
                junk.add_(1)
            for k, (a, b) in enumerate(zip(ref, run())):
                bad[k] += int(not torch.equal(a, b))
        print(f"{M}x{N}x{K} direct={direct} balanced={bal}: mismatching launches of {reps} (bf16 / gelu / resid): {bad}")
        bad_total += sum(bad)
L.cpx_gemm_set_direct(1); L.cpx_gemm_set_balanced(1)
print("RACE SCREEN", "clean" if bad_total == 0 else f"FAILED ({bad_total})")
