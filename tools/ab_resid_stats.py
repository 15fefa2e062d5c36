"""proj / mlp.lin2 shapes with the row-statistics producer epilogue (timing, one build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import ops
dev = torch.device("cuda:0")
M = 32768
g = torch.Generator().manual_seed(0)
for name, N, K in (("proj", 1024, 1024), ("fc2", 1024, 4096)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    out = {}
    for mode in (False, True):
        ts = []
        for rnd in range(4):
            for _ in range(3): ops.gemm_ln(A, W, "resid", b, R, want_stats=mode)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm_ln(A, W, "resid", b, R, want_stats=mode)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        out["stats" if mode else "plain"] = "min %.1f med %.1f" % (min(ts), sorted(ts)[2])
    print(name, out)
