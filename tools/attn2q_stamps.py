"""In-kernel s_memtime stamps of the round-4 attention kernel (k_attention2q<.., DBG=true>, debug build): cycles per
(64 queries x 32 keys) tile of one wave, by segment, at 2 and at 32 sub-tiles (idle chip / full chip)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
names = ["vmcnt wait + barrier", "DMA requests + 10 LDS reads (issue + landing)", "half step A (10 MFMA | softmax chain 0)",
         "half step B (10 MFMA | softmax chain 1)", "seam (G tables, ring wait, pipeline fill)", "whole kernel / items", "epilogue (normalise + stores)"]
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.cpx_attention2q_set_ablation(abl)
print(f'ablation {abl} (0 = the kernel, 1 = without MFMAs, 2 = without the vector stream)')
for nS in (32,):
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    L.cpx_attention_set_variant(2)
    ops.attention(qkv, rel, rel)
    vt = torch.empty((nS * 1024, 1024), dtype=torch.bfloat16, device=dev); out = torch.empty_like(vt)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.cpx_attention(0, qkv.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, vt.data_ptr(), out.data_ptr(), st))   # fills vt
    dbg = torch.zeros((nS * 64, 4, 8), dtype=torch.int32, device=dev)
    for _ in range(3):
        _lib.check(L.cpx_attention2q_debug(qkv.data_ptr(), vt.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, out.data_ptr(), dbg.data_ptr(), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(L.cpx_attention2q_debug(qkv.data_ptr(), vt.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, out.data_ptr(), dbg.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
    print(f"   wall time of the stamped launch: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
    d = dbg.cpu().numpy().astype(np.int64).reshape(-1, 8)
    d = d[d[:, 7] > 0]                 # only the first min(#CUs, 64 nS) workgroups exist
    items = d[:, 7:8]
    med, p90 = np.median(d / items, 0), np.percentile(d / items, 90, 0)
    print(f"nS={nS}: per-wave cycles (s_memtime) PER ITEM (256 queries x 1024 keys), median over {len(d)} waves of {int(np.median(items))} items each")
    for i, n in enumerate(names):
        per = f"per tile {med[i] / 32:7.0f} (p90 {p90[i] / 32:7.0f})" if i < 4 else ""
        print(f"   {n:48s} {med[i]:9.0f}   {per}")
    print(f"   sum of the four loop segments per tile: {med[:4].sum() / 32:.0f}; whole item / 32: {med[5] / 32:.0f}")
