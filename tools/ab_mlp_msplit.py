"""The MLP of one layer (mlp.lin1 -> mlp.lin2, in-engine configuration) over all 32 768 token rows at once against two passes over 16 384 rows
each (the 134 MB hidden half then stays in the 256 MB Infinity Cache between its producer and its consumer): us per layer-MLP, interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M, C, H = 32768, 1024, 4096
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
W1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(torch.bfloat16).to(dev); b1 = torch.randn(H, generator=g).to(dev)
W2 = (torch.randn(C, H, generator=g) / H ** 0.5).to(torch.bfloat16).to(dev); b2 = torch.randn(C, generator=g).to(dev)
stats = ops.row_stats(x); colsum = W1.float().sum(1).contiguous()
hid = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
out = x.clone(); st2 = torch.zeros((M, 4, 2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
def mlp(m0, m):
    o = m0 * C * 2; oh = m0 * H * 2
    _lib.check(L.cpx_gemm_ln(x.data_ptr() + o, W1.data_ptr(), m, H, C, ops.EPI["gelu"], b1.data_ptr(), None, hid.data_ptr() + (oh if SPLIT_BUF else 0), H,
                             stats.data_ptr() + m0 * 32, colsum.data_ptr(), None, s))
    _lib.check(L.cpx_gemm_ln(hid.data_ptr() + (oh if SPLIT_BUF else 0), W2.data_ptr(), m, C, H, ops.EPI["resid"], b2.data_ptr(), out.data_ptr() + o, out.data_ptr() + o, C,
                             None, None, st2.data_ptr() + m0 * 32, s))
def whole(): mlp(0, M)
def halves(): mlp(0, M // 2); mlp(M // 2, M // 2)
def quarters():
    for q in range(4): mlp(q * (M // 4), M // 4)
def timeit(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for SPLIT_BUF in (True, False):
    res = {"whole": [], "halves": []}
    for _ in range(2): whole(); halves()
    for rnd in range(6):
        for k, f in ((("whole", whole), ("halves", halves)) if rnd % 2 == 0 else (("halves", halves), ("whole", whole))):
            res[k].append(timeit(f))
    print(("each part its own rows of the hidden buffer" if SPLIT_BUF else "every part re-uses the FIRST rows of the hidden buffer") + ": " +
          "   ".join(f"{k} {np.median(v):7.1f} us" for k, v in res.items()), flush=True)
