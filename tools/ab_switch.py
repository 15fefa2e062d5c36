"""In-process interleaved A/B of a private debug switch on the whole engine step.
usage: python tools/ab_switch.py cpx_gemm_set_reverse [cpx_other_switch ...]      (values 0 and 1)
       python tools/ab_switch.py cpx_gemm_set_dbg=32,48                            (two explicit values; restored to AB_RESTORE)"""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, engine, synth
dev = torch.device("cuda:0"); L = _lib.lib()
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
    out = eng.result(prev)
    return out
steps(3)
for name in sys.argv[1:]:
    vals = (0, 1)
    if "=" in name:
        name, vs = name.split("=")
        vals = tuple(int(x) for x in vs.split(","))
    fn = getattr(L, name)
    res = {v: [] for v in vals}; outs = {}
    for rnd in range(5):
        for v in vals:
            fn(v)
            o = steps(2); torch.cuda.synchronize(); t = time.perf_counter(); o = steps(20); torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t) / 20 * 1e3)
            outs[v] = (o.dP.clone(), o.masks.clone())
    fn(int(os.environ.get('AB_RESTORE', '1')))
    print(name, "bitwise equal outputs:", torch.equal(outs[vals[0]][0], outs[vals[1]][0]) and torch.equal(outs[vals[0]][1], outs[vals[1]][1]))
    for v in vals:
        print(f"  {name}({v}) engine ms/step: min {min(res[v]):.3f} median {sorted(res[v])[2]:.3f}  {[round(x, 2) for x in res[v]]}")
