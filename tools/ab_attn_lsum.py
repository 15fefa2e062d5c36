"""Production attention kernel (k_attention4p, peeled tail) against its two template experiments in the debug build: LSUM (row sums by an
all-ones MFMA, cpx_attention_set_lsum(1)) and ERD (early fragment reads, cpx_attention_set_lsum(2)); one process, alternating order."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
nS = 32
g = torch.Generator().manual_seed(0)
qkv = torch.randn(nS * 1024, 3072, generator=g).to(torch.bfloat16).to(dev)
rel = lambda: torch.cat([(torch.randn(63, 64, generator=g) * 0.2 * 8), torch.zeros(1, 64)]).to(torch.bfloat16).to(dev)
rh, rw = rel(), rel()
t = {0: [], 1: [], 2: [], 3: []}; outs = {}
for r in range(6):
    for v in ((0, 1, 2, 3) if r % 2 == 0 else (3, 2, 1, 0)):
        L.cpx_attention_set_lsum(v)
        for _ in range(3): o = ops.attention(qkv, rh, rw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): o = ops.attention(qkv, rh, rw)
        e1.record(); torch.cuda.synchronize()
        t[v].append(e0.elapsed_time(e1) / 20 * 1e3); outs[v] = o
L.cpx_attention_set_lsum(0)
ref64 = None
for v, name in ((0, "production"), (1, "LSUM"), (2, "ERD"), (3, "NV (no vote)")):
    d = (outs[v].float() - outs[0].float()).abs().max().item()
    print(f"{name:12s} {np.median(t[v]):7.1f} us (min {min(t[v]):7.1f})   ratio {np.median(t[v]) / np.median(t[0]):6.3f}   max |diff| vs production {d:.3e}")
