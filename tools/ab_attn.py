"""Attention kernel timing + correctness vs a saved reference output (for A/B across builds on one box,
run the OLD build first with `save`, then the new one with `check`)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import ops
dev = torch.device("cuda:0")
nS = 32
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(dev)
rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(dev)
out = ops.attention(qkv, rel, rel)
ts = []
for rnd in range(5):
    for _ in range(3): ops.attention(qkv, rel, rel)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.attention(qkv, rel, rel)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print("attention(+vT) us: min %.1f median %.1f" % (min(ts), sorted(ts)[2]), "checksum", float(out.float().sum()), float(out.float().abs().sum()))
