"""A/B of two k_attention4p experiments against production (cpx_attention_set_lsum: 0 production, 1 = softmax denominators
by an all-ones MFMA instead of the vector add tree, 2 = early fragment reads -- K(t+1) and V(t) requested behind the barrier,
the softmax running under their LDS latency): interleaved rounds in one process, error against a float64 reference,
repeatability under concurrent load, bitwise comparison with production."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
nS = 32
g = torch.Generator().manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(dt).to(dev)
    rel = (torch.randn(64, 64, generator=g) * 0.8).to(dt).to(dev)
    rel[63] = 0
    q, k, v = qkv[:1024].double().reshape(1024, 3, 16, 64).permute(1, 2, 0, 3)
    idx = (torch.arange(32)[:, None] - torch.arange(32)[None, :] + 31).to(dev)
    R = rel.double()[idx] / 8
    qhw = q.reshape(16, 32, 32, 64)
    bias = (torch.einsum("nhwc,hkc->nhwk", qhw, R)[..., :, None] + torch.einsum("nhwc,wkc->nhwk", qhw, R)[..., None, :]).reshape(16, 1024, 1024)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias, -1) @ v).transpose(0, 1).reshape(1024, 1024)
    outs, t = {}, {0: [], 1: [], 2: []}
    noise = torch.empty((8192, 8192), device=dev); side = torch.cuda.Stream(dev)
    for v_ in (0, 1, 2):
        L.cpx_attention_set_lsum(v_)
        outs[v_] = ops.attention(qkv, rel, rel)
        stable = True
        for i in range(12):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    noise.normal_()
            stable &= torch.equal(ops.attention(qkv, rel, rel), outs[v_])
        side.synchronize()
        err = float((outs[v_][:1024].double() - ref).norm() / ref.norm())
        print(f"{str(dt):16s} variant={v_}: rel-L2 vs float64 {err:.3e}   repeatable {stable}   bitwise equal to production {torch.equal(outs[v_], outs[0])}")
    for rnd in range(6):
        for v_ in (0, 1, 2):
            L.cpx_attention_set_lsum(v_)
            for _ in range(3): ops.attention(qkv, rel, rel)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.attention(qkv, rel, rel)
            e1.record(); torch.cuda.synchronize()
            t[v_].append(e0.elapsed_time(e1) / 20 * 1e3)
    for v_ in (0, 1, 2):
        print(f"{str(dt):16s} variant={v_}: attention + V transpose median {np.median(t[v_]):7.1f} us  min {min(t[v_]):7.1f}")
L.cpx_attention_set_lsum(0)
