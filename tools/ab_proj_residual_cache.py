"""attn.proj (bf16, residual in place + row statistics, M = 32 768) with its operands in different cache states: after 512 MB of unrelated traffic has pushed
everything out of the 256 MB Infinity Cache ("cold"), with only the residual rows x touched again, only the attention output (A operand), or both."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M, N, K = 32768, 1024, 1024
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
x = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
stats = torch.zeros((M, 4, 2), dtype=torch.float32, device=dev)
big = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def proj():
    _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["resid"], b.data_ptr(), x.data_ptr(), x.data_ptr(), N, None, None, stats.data_ptr(), st))
def run(touch_x, touch_a, n=12):
    ts = []
    for _ in range(n):
        big.add_(1)                                   # 1.5 GB of read + write traffic: evicts the Infinity Cache
        if touch_a: A.view(torch.int16).bitwise_and_(-1)     # read + write back the same bits: the lines are the most recent
        if touch_x: x.view(torch.int16).bitwise_and_(-1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); proj(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))
for _ in range(3): proj()
for name, tx, ta in (("cold (nothing resident)", 0, 0), ("residual x resident", 1, 0), ("attention output (A) resident", 0, 1), ("both resident", 1, 1), ("cold again", 0, 0)):
    print(f"{name:32s} {run(tx, ta):7.1f} us", flush=True)
# (round 5 also built the residual epilogue with the residual rows of BOTH 128-row halves requested up front -- 16 loads in flight per lane instead of
# 8 -- and measured it in the state the engine runs attn.proj in, A resident / x cold: 83.0 - 83.8 against 78.6 - 79.3 us, engine step 21.89 against
# 21.82 ms: slower, not kept; profiles/r05_ab_proj_residual_cache.txt)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): proj()
e1.record(); torch.cuda.synchronize()
print(f"{'back to back (all resident)':32s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
