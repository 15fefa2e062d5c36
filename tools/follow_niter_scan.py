"""Fixed and per-step cost of a11 (k_prep_flow + k_follow) by niter, LDS window off / on (debug library), on the bench's analytic fields."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops, synth
dev = torch.device("cuda:0"); L = _lib.lib()
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
dP = torch.from_numpy(np.stack([a[0] for a in f])).to(dev); cp = torch.from_numpy(np.stack([a[1] for a in f])).to(dev)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for ee in (1, 0):
    L.cpx_follow_set_early_exit(ee)
    for v in (0, 2, 9, 11, 1, 3, 7):
        L.cpx_follow_set_lds_window(v)
        print("early_exit", ee, "lds window", v, " ".join("niter %d: %.1f us" % (n, timed(lambda: ops.follow_flows(dP, cp, niter=n))) for n in (0, 1, 25, 50, 100, 200, 400)))
L.cpx_follow_set_early_exit(1); L.cpx_follow_set_lds_window(3)
