import os
os.environ["CLASSPOSE_HIP_DEBUG"]="1"
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from classpose_amd import _lib, ops
dev=torch.device("cuda:0"); L=_lib.lib()
M,N,K=32768,3072,1024
g=torch.Generator().manual_seed(0)
A=torch.randn(M,K,generator=g).to(torch.bfloat16).to(dev)
W=(torch.randn(N,K,generator=g)/K**0.5).to(torch.bfloat16).to(dev)
b=torch.randn(N,generator=g).to(dev)
stats=ops.row_stats(A); cs=W.float().sum(1).contiguous()
out=torch.empty((M,N),dtype=torch.bfloat16,device=dev)
vt=torch.empty((M//1024,16,64,1024),dtype=torch.bfloat16,device=dev)
st=torch.cuda.current_stream().cuda_stream
def run(epi):
    _lib.check(L.cpx_gemm_ln(A.data_ptr(),W.data_ptr(),M,N,K,ops.EPI[epi],b.data_ptr(),vt.data_ptr() if epi=="qkv" else None,out.data_ptr(),N,stats.data_ptr(),cs.data_ptr(),None,st))
t={"qkv":[], "bf16":[]}
for rnd in range(6):
    for epi in ("qkv","bf16"):
        for _ in range(3): run(epi)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(epi)
        e1.record(); torch.cuda.synchronize(); t[epi].append(e0.elapsed_time(e1)/20*1e3)
for k,v in t.items(): print(f"N=3072 K=1024 LN-folded, epilogue {k:5s}: median {np.median(v):7.1f} us min {min(v):7.1f}")
