"""Timing-only ablations of post-processing stages inside the engine step (interleaved)."""
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
def steps(n, records=True):
    prev = None
    for _ in range(n):
        sid = eng.submit(tiles, inject=inj, records=records)
        if prev is not None: eng.result(prev)
        prev = sid
    eng.result(prev)
variants = {"full": dict(niter=200, flow_thr=0.4, rec=True), "niter=0 (no Euler loop)": dict(niter=0, flow_thr=0.4, rec=True),
            "no flow-error filter": dict(niter=200, flow_thr=0.0, rec=True), "no records": dict(niter=200, flow_thr=0.4, rec=False)}
res = {k: [] for k in variants}
steps(3)
for rnd in range(4):
    for k, v in variants.items():
        eng.niter, eng.flow_thr = v["niter"], v["flow_thr"]
        steps(2, v["rec"]); torch.cuda.synchronize(); t = time.perf_counter(); steps(20, v["rec"]); torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t) / 20 * 1e3)
for k, v in res.items(): print(f"{k:28s} ms/step: min {min(v):.3f} median {sorted(v)[2]:.3f}")
