"""Run the attention kernel alone (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import ops
dev = torch.device("cuda:0")
nS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(0)
qkv = torch.randn(nS * 1024, 3072, generator=g).to(torch.bfloat16).to(dev)
rel = (torch.randn(64, 64, generator=g) * 0.3 * 8).to(torch.bfloat16).to(dev)
for _ in range(5):
    out = ops.attention(qkv, rel, rel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = ops.attention(qkv, rel, rel)
e1.record(); torch.cuda.synchronize()
print("attention+vT avg us", e0.elapsed_time(e1) / 10 * 1e3)
