"""In-kernel s_memtime stamps of the attention loop (diagnostic build k_attention<.., DBG=true>)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
names = ["QK: gh+K reads+4 MFMA", "max/swap/vote", "fma+exp+pack", "PV: V reads+6 MFMA", "prefetch wait+LDS write", "barrier", "-", "prologue", "TOTAL"]
for nS in (2, 32):
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    vt = torch.empty((nS * 1024, 1024), dtype=torch.bfloat16, device=dev); out = torch.empty_like(vt)
    dbg = torch.zeros((nS * 128, 4, 9), dtype=torch.int32, device=dev)
    for _ in range(2):
        _lib.check(L.cpx_attention_debug(qkv.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, vt.data_ptr(), out.data_ptr(), dbg.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.int64).reshape(-1, 9)
    med = np.median(d, 0)
    print(f"nS={nS}: median per-wave cycles (s_memtime ticks); loop segments are sums over 32 tiles")
    for n, v in zip(names, med):
        if n != "-": print(f"   {n:28s} {v:9.0f}   per tile {v/32:7.0f}")
