import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops, synth
dev = torch.device("cuda:0"); L = _lib.lib()
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
dP = torch.from_numpy(np.stack([a[0] for a in f])).to(dev); cp = torch.from_numpy(np.stack([a[1] for a in f])).to(dev)
g = torch.Generator().manual_seed(0)
dPr = (torch.randn(8, 2, 256, 256, generator=g) * 2).to(dev); cpr = torch.randn(8, 256, 256, generator=g).to(dev)
import scipy.ndimage as ndi
sm = torch.from_numpy(np.stack([[ndi.gaussian_filter(x, 4) * 20 for x in t] for t in dPr.cpu().numpy()])).to(dev)
for name, a, b in (("analytic", dP, cp), ("random", dPr, cpr), ("smooth random", sm, cpr)):
    for v in (0, 1):
        L.cpx_follow_set_early_exit(v)
        for _ in range(2): out = ops.follow_flows(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = ops.follow_flows(a, b)
        e1.record(); torch.cuda.synchronize()
        print(name, "early_exit", v, "%.1f us" % (e0.elapsed_time(e1) / 10 * 1e3))
