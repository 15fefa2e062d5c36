"""Round-5 review item 7: "try the per-page MALL no-alloc attribute for the qkv output buffer".

HIP 7.2 exposes no per-allocation Infinity-Cache (MALL) policy; the one cache attribute an allocation can carry is hipDeviceMallocUncached
(hipExtMallocWithFlags: MTYPE UC -- the pages bypass the L2 as well as the MALL).  This A/B puts the attention kernel's input (the qkv
projection's output: q, k rows and the V^T workspace, 268 MB for 32 sub-tiles) in such an allocation and times attention on it against an ordinary one,
interleaved rounds, outputs compared bit for bit.  The eight query-block workgroups of a head share its K / V^T tiles through the L2, so an
attribute that also switches the L2 off is expected to cost the kernel more than the Infinity Cache space it frees could return to attn.proj."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from classpose_amd import _lib, ops

dev = torch.device("cuda:0")
L = _lib.lib()
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
HIP_DEVICE_MALLOC_UNCACHED = 0x3
nS = 32
M = nS * 1024
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(M, 3072, generator=g) * 0.5).to(torch.bfloat16).to(dev)
rel = [(torch.randn(64, 64, generator=g) * 0.1).to(torch.bfloat16).to(dev) for _ in range(2)]
for r in rel:
    r[63] = 0
vt = torch.empty((M, 1024), dtype=torch.bfloat16, device=dev)
out_a = torch.empty((M, 1024), dtype=torch.bfloat16, device=dev)
out_b = torch.empty_like(out_a)
p_qkv, p_vt = C.c_void_p(), C.c_void_p()
assert hip.hipExtMallocWithFlags(C.byref(p_qkv), qkv.numel() * 2, HIP_DEVICE_MALLOC_UNCACHED) == 0
assert hip.hipExtMallocWithFlags(C.byref(p_vt), vt.numel() * 2, HIP_DEVICE_MALLOC_UNCACHED) == 0
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(p_qkv, qkv.data_ptr(), qkv.numel() * 2, 3) == 0
st = torch.cuda.current_stream().cuda_stream


def run(q, v, o):
    _lib.check(L.cpx_attention(_lib.DT_BF16, q, rel[0].data_ptr(), rel[1].data_ptr(), nS, v, o, st), "attention")


def timeit(q, v, o, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(q, v, o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


A = (qkv.data_ptr(), vt.data_ptr(), out_a.data_ptr())
B = (p_qkv.value, p_vt.value, out_b.data_ptr())
for _ in range(3):
    run(*A); run(*B)
torch.cuda.synchronize()
res = {"ordinary": [], "uncached": []}
for rnd in range(6):
    for name, a in ((("ordinary", A), ("uncached", B)) if rnd % 2 == 0 else (("uncached", B), ("ordinary", A))):
        res[name].append(timeit(*a))
print(f"attention, {nS} sub-tiles (us per launch, median / min over 6 interleaved rounds of 20):")
for k, v in res.items():
    print(f"  qkv + V^T in {k:9s} memory  {np.median(v):7.1f} / {min(v):7.1f}")
print(f"  uncached / ordinary = {np.median(res['uncached']) / np.median(res['ordinary']):.3f}   outputs bitwise equal: {bool(torch.equal(out_a, out_b))}")
hip.hipFree(p_qkv); hip.hipFree(p_vt)
