"""Post-processing alone (compute_masks + records on injected fields), for rocprofv3 --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import ops, synth
dev = torch.device("cuda:0")
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
dP, cp, lg = (torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
for _ in range(12):
    out = ops.compute_masks(dP, cp, lg)
torch.cuda.synchronize()
print("ok", [int(x) for x in out[2][:3]] if len(out) > 2 else "")
