"""A/B of the attention kernels in one process (interleaved rounds): 4-wave register-ring kernel (variant 0), 8-wave
ping-pong (variant 1), 4-wave LDS-DMA ring + software-pipelined S (variant 2); max |difference| of the outputs and both against a float64 reference on 1 sub-tile."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
nS = 32
qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
relh = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev); relh[63] = 0
relw = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev); relw[63] = 0
outs = {}
for v in (2, 1, 0):
    L.cpx_attention_set_variant(v)
    outs[v] = ops.attention(qkv, relh, relw).float()
for v in (1, 2):
    print(f"max |variant {v} - variant 0|:", float((outs[v] - outs[0]).abs().max()), " rel-L2:", float((outs[v] - outs[0]).norm() / outs[0].norm()))
# float64 reference on sub-tile 0, head 3
q, k, v_ = qkv[:1024].double().reshape(1024, 3, 16, 64).permute(1, 2, 0, 3)
idx = (torch.arange(32)[:, None] - torch.arange(32)[None, :] + 31).to(dev)
Rh, Rw = relh.double()[idx] / 8, relw.double()[idx] / 8
qhw = q.reshape(16, 32, 32, 64)
bias = (torch.einsum("nhwc,hkc->nhwk", qhw, Rh)[..., :, None] + torch.einsum("nhwc,wkc->nhwk", qhw, Rw)[..., None, :]).reshape(16, 1024, 1024)
ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias, -1) @ v_).transpose(0, 1).reshape(1024, 1024)
for v in (2, 1, 0):
    d = outs[v][:1024].double() - ref
    print(f"variant {v}: rel-L2 vs float64 {float(d.norm() / ref.norm()):.3e}, max-abs {float(d.abs().max()):.3e}")
vt = torch.empty((nS * 1024, 1024), dtype=torch.bfloat16, device=dev); out = torch.empty_like(vt)
st = torch.cuda.current_stream().cuda_stream
res = {0: [], 1: [], 2: []}
for rnd in range(6):
    for v in (2, 1, 0):
        L.cpx_attention_set_variant(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(25):
            if i == 5: e0.record()
            _lib.check(L.cpx_attention_relpos(qkv.data_ptr(), relh.data_ptr(), relw.data_ptr(), nS, vt.data_ptr(), out.data_ptr(), st))
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20 * 1e3)
L.cpx_attention_set_variant(0)
fl = nS * (4.0 * 1024 * 1024 * 64 * 16 + 4.0 * 16 * 1024 * 64 * 64)
for v in (2, 1, 0):
    m = float(np.median(res[v]))
    print(f"variant {v}: median {m:.1f} us (incl. the V transpose kernel), min {min(res[v]):.1f}  -> {fl / m / 1e6:.0f} TFLOP/s")
