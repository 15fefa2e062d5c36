"""Cost of the folded-LayerNorm consumer epilogue (qkv and mlp.lin1 shapes), one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0")
M, K = 32768, 1024
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
st = ops.row_stats(A)
vt = torch.empty((M, 1024), dtype=torch.bfloat16, device=dev)
for name, N, epi in (("qkv", 3072, "qkv"), ("fc1", 4096, "gelu")):
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    cs = W.float().sum(1).contiguous()
    aux = vt if epi == "qkv" else None
    res = {"plain": [], "ln": []}
    for rnd in range(4):
        for mode in ("plain", "ln"):
            kw = dict(ln_stats=st, ln_colsum=cs) if mode == "ln" else {}
            for _ in range(3): ops.gemm_ln(A, W, epi, b, aux, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm_ln(A, W, epi, b, aux, **kw)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
    print(name, {k: "min %.1f med %.1f" % (min(v), sorted(v)[2]) for k, v in res.items()})
