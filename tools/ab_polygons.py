"""In-process A/B: engine pipeline with and without device polygonisation (interleaved rounds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import engine, synth
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
eng = engine.Engine(w, 256, batch_tiles=8)
org = [(224 * i, 0) for i in range(8)]
tiles = torch.from_numpy(np.stack([synth.render_region(1234, x, y, 256, 256) for x, y in org])).to(dev)
f = [synth.analytic_fields(1234, x, y, 256, 256, 7) for x, y in org]
inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
def steps(n, poly):
    prev = None
    for _ in range(n):
        sid = eng.submit(tiles, inject=inj, records=True, polygons=(1.0, org) if poly else None)
        if prev is not None: eng.result(prev)
        prev = sid
    eng.result(prev)
res = {False: [], True: []}
steps(3, True); steps(3, False); torch.cuda.synchronize()
for rnd in range(5):
    for poly in (False, True):
        torch.cuda.synchronize(); t = time.perf_counter(); steps(20, poly); torch.cuda.synchronize()
        res[poly].append((time.perf_counter() - t) / 20 * 1e3)
for k, v in res.items():
    print("polygons" if k else "records only", "ms/step: min %.3f median %.3f" % (min(v), sorted(v)[len(v) // 2]), [round(x, 2) for x in v])
