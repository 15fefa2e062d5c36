"""In-process A/B of the attention kernel's XCD-aware workgroup order (interleaved rounds) + full-engine A/B."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops, engine, synth
dev = torch.device("cuda:0"); L = _lib.lib()
nS = 32
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
L.cpx_attention_set_xcd_order(0); ref = ops.attention(qkv, rel, rel).clone()
L.cpx_attention_set_xcd_order(1); new = ops.attention(qkv, rel, rel)
print("bitwise equal:", torch.equal(ref, new))
res = {0: [], 1: []}
for rnd in range(5):
    for v in (0, 1):
        L.cpx_attention_set_xcd_order(v)
        for _ in range(3): ops.attention(qkv, rel, rel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.attention(qkv, rel, rel)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 20 * 1e3)
for v in (0, 1):
    print("xcd order", v, "attention(+vT) us: min %.1f median %.1f" % (min(res[v]), sorted(res[v])[2]))
# whole engine
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
eng = engine.Engine(w, 256, batch_tiles=8)
tiles = torch.from_numpy(np.stack([synth.render_region(1234, 224 * i, 0, 256, 256) for i in range(8)])).to(dev)
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
def steps(n):
    prev = None
    for _ in range(n):
        sid = eng.submit(tiles, inject=inj, records=True)
        if prev is not None: eng.result(prev)
        prev = sid
    eng.result(prev)
res = {0: [], 1: []}
steps(3)
for rnd in range(4):
    for v in (0, 1):
        L.cpx_attention_set_xcd_order(v)
        steps(2); torch.cuda.synchronize(); t = time.perf_counter(); steps(20); torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t) / 20 * 1e3)
for v in (0, 1):
    print("xcd order", v, "engine ms/step: min %.3f median %.3f" % (min(res[v]), sorted(res[v])[len(res[v]) // 2]))
L.cpx_attention_set_xcd_order(1)
