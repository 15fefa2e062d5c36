"""Attention kernel time per sub-tile against the number of sub-tiles in the launch (workgroups per CU = n_subtiles x 16 heads x 8 query blocks / 256):
does the last, partly filled round of workgroups (3 per CU resident; 32 sub-tiles = 16 per CU = 5.33 rounds) cost anything?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rel_h = (torch.randn(64, 64, generator=g) * 0.1).to(torch.bfloat16).to(dev); rel_w = (torch.randn(64, 64, generator=g) * 0.1).to(torch.bfloat16).to(dev)
for nS in (6, 12, 18, 24, 27, 30, 32, 33, 36, 42, 48, 64):
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    for _ in range(3): out = ops.attention(qkv, rel_h, rel_w)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = ops.attention(qkv, rel_h, rel_w)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(ts)[2]
    print(f"n_subtiles {nS:3d}  workgroups per CU {nS * 128 / 256:5.1f}  rounds of 3 per CU {nS * 128 / 768:5.2f}   {t:7.1f} us  {t / nS:6.2f} us per sub-tile")
