import os
os.environ["CLASSPOSE_HIP_DEBUG"]="1"
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from classpose_amd import _lib, ops
dev=torch.device("cuda:0"); L=_lib.lib()
g=torch.Generator().manual_seed(0); nS=32
qkv=(torch.randn(nS*1024,3072,generator=g)*0.7).to(torch.bfloat16).to(dev)
rel=(torch.randn(64,64,generator=g)*0.8).to(torch.bfloat16).to(dev)
vt=torch.empty((nS*1024,1024),dtype=torch.bfloat16,device=dev); out=torch.empty_like(vt)
dbg=torch.zeros((nS*128,4,6),dtype=torch.int32,device=dev)
st=torch.cuda.current_stream().cuda_stream
def prod(): ops.attention(qkv,rel,rel)
def stamped(): _lib.check(L.cpx_attention4_debug(qkv.data_ptr(),rel.data_ptr(),rel.data_ptr(),nS,vt.data_ptr(),out.data_ptr(),dbg.data_ptr(),st))
ref=ops.attention(qkv,rel,rel); stamped(); torch.cuda.synchronize()
print("stamped output equals production:", torch.equal(ref,out))
t={"prod":[], "stamped":[]}
for rnd in range(6):
    for name,fn in (("prod",prod),("stamped",stamped)):
        for _ in range(3): fn()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); t[name].append(e0.elapsed_time(e1)/20*1e3)
for k,v in t.items(): print(k, "median %.1f us min %.1f"%(np.median(v),min(v)))
