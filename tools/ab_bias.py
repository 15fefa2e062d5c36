import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0")
M, N, K = 32768, 3072, 1024
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
res = {"bias": [], "nobias": []}
for rnd in range(5):
    for k, bb in (("bias", b), ("nobias", None)):
        for _ in range(3): ops.gemm(A, W, "bf16", bb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(A, W, "bf16", bb)
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
for k, v in res.items(): print(k, "min %.1f med %.1f" % (min(v), sorted(v)[2]))
