import os
os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M, N, K = 8192, 2048, 1024
g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
bias = torch.randn(N, generator=g).to(dev)
stats = ops.row_stats(A); colsum = W.float().sum(1).contiguous()
run = lambda: ops.gemm_ln(A, W, "gelu", bias, None, ln_stats=stats, ln_colsum=colsum)
L.cpx_gemm_set_direct(0); L.cpx_gemm_set_balanced(0); ref = run()
for d, b in ((1, 0), (2, 0), (0, 2), (2, 2)):
    L.cpx_gemm_set_direct(d); L.cpx_gemm_set_balanced(b)
    o = run()
    df = (o.float() - ref.float()).abs()
    bad = df > 0
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print(f"direct={d} bal={b}: {int(bad.sum())} differ, max {float(df.max()):.3e}; rows {rows[:6].tolist()}..{rows[-3:].tolist() if len(rows) else []} (n={len(rows)}), cols n={len(cols)} first {cols[:8].tolist()}")
    if len(rows):
        r = rows % 256; c = cols % 256
        print("   row offsets in tile (unique, first 20):", torch.unique(r)[:20].tolist(), " col offsets:", torch.unique(c)[:20].tolist())
L.cpx_gemm_set_direct(1); L.cpx_gemm_set_balanced(1)
