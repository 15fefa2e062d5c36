import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from classpose_amd import _lib, ops, synth
dev = torch.device("cuda:0"); L = _lib.lib()
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
dP = torch.from_numpy(np.stack([a[0] for a in f])).to(dev); cp = torch.from_numpy(np.stack([a[1] for a in f])).to(dev)
A = torch.randn(8192, 4096, device=dev, dtype=torch.bfloat16); B = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
def t(niter, ee, warm_gemm=False, reps=10):
    L.cpx_follow_set_early_exit(ee)
    for _ in range(2): ops.follow_flows(dP, cp, niter=niter)
    if warm_gemm:
        for _ in range(30): torch.mm(A, B)
    torch.cuda.synchronize() if not warm_gemm else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.follow_flows(dP, cp, niter=niter)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for ee in (1, 0):
    print("early_exit", ee, {n: round(t(n, ee), 1) for n in (0, 25, 50, 100, 200, 400)})
print("after a GEMM burst (clocks up), 200 iters:", round(t(200, 1, True), 1), round(t(200, 0, True), 1))
fg = float((cp > 0).float().mean())
print("foreground fraction", fg)
L.cpx_follow_set_early_exit(1)
