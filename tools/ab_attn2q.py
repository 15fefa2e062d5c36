"""Round-4 attention kernel (variant 3, k_attention2q: one wave per SIMD, two query rows per wave) against the production
kernel of rounds 2-3 (variant 2, k_attention4p): float64 error of both, repeatability, the rescale path, one-process timing."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
VARIANTS = (2, 3, 4, 5, 6)   # 2 production (k_attention4p), 3 k_attention2q with the LDS ring, 4 with fragments straight from global memory,
                          # 6 = fragments requested TWO tiles ahead into four rotating AGPR sets;
                          # 5 = 4 without the per-half-step overflow vote (per-item check + exact re-run; bf16 only, fp16 falls back to 4)
nS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def rel_l2(a, b): return float((a - b).norm() / b.norm())
def ref64(qkv, rel, s0=0):
    q, k, v = qkv[s0 * 1024:(s0 + 1) * 1024].double().reshape(1024, 3, 16, 64).permute(1, 2, 0, 3)
    idx = (torch.arange(32)[:, None] - torch.arange(32)[None, :] + 31).to(dev)
    R = rel.double()[idx] / 8
    qhw = q.reshape(16, 32, 32, 64)
    bias = (torch.einsum("nhwc,hkc->nhwk", qhw, R)[..., :, None] + torch.einsum("nhwc,wkc->nhwk", qhw, R)[..., None, :]).reshape(16, 1024, 1024)
    return (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias, -1) @ v).transpose(0, 1).reshape(1024, 1024)
for dt in (torch.bfloat16, torch.float16):
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(dt).to(dev)
    rel = (torch.randn(64, 64, generator=g) * 0.8).to(dt).to(dev); rel[63] = 0
    outs = {}
    for v in VARIANTS:
        L.cpx_attention_set_variant(v)
        outs[v] = ops.attention(qkv, rel, rel)
        for i in range(10):
            assert torch.equal(ops.attention(qkv, rel, rel), outs[v]), ("not repeatable", v, i)
    torch.cuda.synchronize()
    for s0 in (0, nS - 1):
        r = ref64(qkv, rel, s0)
        print(f"{dt}: sub-tile {s0}: rel-L2 vs float64  " + "   ".join(f"variant {v} {rel_l2(outs[v][s0*1024:(s0+1)*1024].double(), r):.3e}" for v in VARIANTS) +
              f"   4 == 3 bitwise: {bool(torch.equal(outs[4], outs[3]))}  finite {bool(torch.isfinite(outs[4].float()).all())}")
    # spiked rows: the rescale path
    q2 = (torch.randn(1024, 3072, generator=g) * 0.1); q2[:, :1024] = 1.0; q2[700, 1024:2048] = 30.0
    q2 = q2.to(dt).to(dev); z = torch.zeros(64, 64, dtype=dt, device=dev)
    r2 = ref64(q2, z)
    for v in VARIANTS:
        L.cpx_attention_set_variant(v)
        o = ops.attention(q2, z, z)
        print(f"{dt}: spiked key: variant {v} max |err| {float((o.double() - r2).abs().max()):.3e}")
# ramp: logits grow with the key index (up to ~60 nats over the 1024 keys), so the running reference is overtaken again
# and again and the exact maximum / rescale path of every chain runs many times
g = torch.Generator().manual_seed(21)
u = torch.randn(16, 64, generator=g); u = u / u.norm(dim=1, keepdim=True)
q3 = torch.randn(1024, 3072, generator=g) * 0.3
q3[:, :1024] += (u * 8).reshape(1, 1024)                                   # q ~ 8 u_head + noise
ramp = (torch.arange(1024).float() / 1024 * 60)[:, None, None]             # logit 0.125 * 8 * ramp = ramp
q3[:, 1024:2048] += (u[None] * ramp).reshape(1024, 1024)
for dt in (torch.bfloat16, torch.float16):
    qq = q3.to(dt).to(dev); z = torch.zeros(64, 64, dtype=dt, device=dev)
    r3 = ref64(qq, z)
    for v in VARIANTS:
        L.cpx_attention_set_variant(v)
        o = ops.attention(qq, z, z)
        print(f"{dt}: ramp (repeated rescale): variant {v} rel-L2 vs float64 {rel_l2(o.double(), r3):.3e}  finite {bool(torch.isfinite(o.float()).all())}")
# timing (bf16), interleaved
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev); rel[63] = 0
vt = torch.empty((nS * 1024, 1024), dtype=torch.bfloat16, device=dev); out = torch.empty_like(vt)
st = torch.cuda.current_stream().cuda_stream
t = {v: [] for v in VARIANTS}
for rnd in range(8):
    for v in VARIANTS:
        L.cpx_attention_set_variant(v)
        for _ in range(3): ops.attention(qkv, rel, rel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.cpx_attention(0, qkv.data_ptr(), rel.data_ptr(), rel.data_ptr(), nS, vt.data_ptr(), out.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        t[v].append(e0.elapsed_time(e1) / 20 * 1e3)
L.cpx_attention_set_variant(2)
for v in VARIANTS:
    m = np.median(t[v])
    print(f"variant {v}: median {m:7.1f} us  min {min(t[v]):7.1f}  (incl. the V transpose kernel)  {4.43e9 * nS / m / 1e6:7.1f} TFLOP/s = {4.43e9 * nS / m / 1e6 / 2500:.3f} of 2.5 PF")
