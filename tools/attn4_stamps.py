"""In-kernel s_memtime stamps of the PRODUCTION attention loop (k_attention4p<.., DBG=true>, debug build): where an
in-order wave spends its cycles per 32-key tile, at 2 and at 32 sub-tiles (idle chip / full chip)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
names = ["vmcnt wait + barrier", "DMA + gh + K reads + wait + 4 QK MFMA (issue)", "softmax vector stream", "V reads + wait + 4 PV MFMA (issue)", "prologue", "whole kernel"]
for nS in (2, 32):
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    vt = torch.empty((nS * 1024, 1024), dtype=torch.bfloat16, device=dev); out = torch.empty_like(vt)
    dbg = torch.zeros((nS * 128, 4, 6), dtype=torch.int32, device=dev)
    for _ in range(3):
        _lib.check(L.cpx_attention4_debug(qkv.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, vt.data_ptr(), out.data_ptr(), dbg.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.int64).reshape(-1, 6)
    med, p90 = np.median(d, 0), np.percentile(d, 90, 0)
    print(f"nS={nS}: per-wave cycles (s_memtime), median over {len(d)} waves; loop segments are sums over 32 tiles")
    for i, n in enumerate(names):
        per = f"per tile {med[i] / 32:7.0f} (p90 {p90[i] / 32:7.0f})" if i < 4 else ""
        print(f"   {n:48s} {med[i]:9.0f}   {per}")
    print(f"   sum of the four loop segments per tile: {med[:4].sum() / 32:.0f}; whole kernel / 32: {med[5] / 32:.0f}")
