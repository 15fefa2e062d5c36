"""A/B of the Euler loop's LDS window (cpx_follow_set_lds_window): 32 x 32-cell foreground segments whose neighbourhood of the flow
field is staged in LDS (1, production) against round 4's row-major segments with every tap from memory (0).
One process, interleaved; a11 alone (cpx_follow_flows: k_prep_flow + k_follow) and the whole fused chain (cpx_compute_masks), on the
bench's analytic fields, on white-noise fields (no chain converges, positions wander out of the window) and on smoothed noise;
at 256 x 256 and at 512 x 512 x 2 tiles; end points / masks compared bit for bit.
usage: python tools/ab_follow_lds.py"""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.ndimage as ndi
from classpose_amd import _lib, ops, synth
dev = torch.device("cuda:0"); L = _lib.lib()


def fields(n, hw):
    f = [synth.analytic_fields(1234, 224 * i, 0, hw, hw, 7) for i in range(n)]
    dP = torch.from_numpy(np.stack([a[0] for a in f])).to(dev); cp = torch.from_numpy(np.stack([a[1] for a in f])).to(dev)
    lg = torch.from_numpy(np.stack([a[2] for a in f])).to(dev)
    g = torch.Generator().manual_seed(0)
    dPr = (torch.randn(n, 2, hw, hw, generator=g) * 2).to(dev); cpr = torch.randn(n, hw, hw, generator=g).to(dev)
    sm = torch.from_numpy(np.stack([[ndi.gaussian_filter(x, 4) * 20 for x in t] for t in dPr.cpu().numpy()])).to(dev)
    return (("analytic (the bench's)", dP, cp, lg), ("white noise", dPr, cpr, lg), ("smoothed noise", sm, cpr, lg))


def timed(fn, n=20):
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


VARIANTS = ((0, "round 4"), (2, "wave test per 4 steps"), (9, "32 x 32 segments"), (11, "32 x 32 segments + wave test"), (3, "... + LDS window (production)"))
for n, hw in ((8, 256), (2, 512)):
    for name, dP, cp, lg in fields(n, hw):
        res = {v: {"a11": [], "chain": []} for v, _ in VARIANTS}; outs = {}
        for rnd in range(3):
            for v, _ in VARIANTS:
                L.cpx_follow_set_lds_window(v)
                t, (pf, fl) = timed(lambda: ops.follow_flows(dP, cp, return_float=True)); res[v]["a11"].append(t)
                t2, (m, cm, nl) = timed(lambda: ops.compute_masks(dP, cp, lg)); res[v]["chain"].append(t2)
                outs[v] = (pf.clone(), fl.clone(), m.clone(), cm.clone(), nl.clone())
        L.cpx_follow_set_lds_window(3)
        med = lambda x: sorted(x)[len(x) // 2]
        print(f"{n} x {hw}^2 {name}: foreground {float((cp > 0).float().mean()):.2f}, {int(outs[3][4].sum())} cells")
        for v, what in VARIANTS:
            same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[v]))
            print(f"    {what:34s} a11 (prep + Euler loop) {med(res[v]['a11']):7.1f} us   fused chain {med(res[v]['chain']):7.1f} us   bitwise equal to round 4: {same}")
