"""Post-processing chain (cpx_compute_masks on one 8-tile batch): eager launches vs replay of a captured HIP graph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import ops, synth
dev = torch.device("cuda:0")
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
dP, cp, lg = (torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
def run():
    return ops.compute_masks(dP, cp, lg)
for _ in range(3): ref = run()
torch.cuda.synchronize()
def timeit(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("eager: %.1f us per batch" % timeit(run))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): run()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph replay equals eager:", all(torch.equal(a, b) for a, b in zip(out, ref)))
print("graph: %.1f us per batch" % timeit(g.replay))
