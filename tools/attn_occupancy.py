"""Attention time vs grid size: are 3 workgroups/CU really co-resident and overlapping?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
nmax = 32
qkv = (torch.randn(nmax * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
vt = torch.empty((nmax * 1024, 1024), dtype=torch.bfloat16, device=dev)
out = torch.empty_like(vt)
import ctypes
for nS in (2, 4, 6, 8, 12, 16, 32):
    def run(n):
        for _ in range(n):
            _lib.check(L.cpx_attention_relpos(qkv.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, vt.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    run(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(20); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"nS {nS:3d}: blocks {128*nS:5d} = {128*nS/256:5.1f} per CU   {t:8.1f} us (incl. V transpose)   {t/(128*nS/256):7.1f} us per block-per-CU")
