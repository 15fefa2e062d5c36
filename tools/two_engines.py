"""Two half-size engines on their own streams against one full-size engine (experiment, round 6): does running two independent 16-sub-tile
forwards side by side (different kernels of the two halves on the chip at the same time: de-phased store bursts, one half's tail under the
other's ramp) beat one 32-sub-tile forward?  Same tiles, flow injection, per-step read-back as in bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import engine, synth
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
T = 256
def batch(n, x0):
    tiles = torch.from_numpy(np.stack([synth.render_region(1234, 224 * (x0 + i), 448, T, T) for i in range(n)])).to(dev)
    f = [synth.analytic_fields(1234, 224 * (x0 + i), 448, T, T, 7) for i in range(n)]
    return tiles, tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
def run(engs, batches, n):
    prev = [None] * len(engs)
    for _ in range(n):
        for i, e in enumerate(engs):
            sid = e.submit(batches[i][0], inject=batches[i][1], records=True)
            if prev[i] is not None:
                int(e.result(prev[i]).nlabels.max())
            prev[i] = sid
    for i, e in enumerate(engs):
        int(e.result(prev[i]).nlabels.max())
    torch.cuda.synchronize()
one = [engine.Engine(w, T, batch_tiles=8)]
b8 = [batch(8, 0)]
two = [engine.Engine(w, T, batch_tiles=4), engine.Engine(w, T, batch_tiles=4)]
b4 = [batch(4, 0), batch(4, 4)]
two8 = [engine.Engine(w, T, batch_tiles=8), engine.Engine(w, T, batch_tiles=8)]
b88 = [batch(8, 0), batch(8, 8)]
res = {"one engine x 8 tiles": [], "two engines x 4 tiles": [], "two engines x 8 tiles": []}
for rnd in range(5):
    for name, engs, bs, tiles_per_round in (("one engine x 8 tiles", one, b8, 8), ("two engines x 4 tiles", two, b4, 8), ("two engines x 8 tiles", two8, b88, 16)):
        run(engs, bs, 3)
        t = time.perf_counter(); n = 20; run(engs, bs, n); dt = time.perf_counter() - t
        res[name].append(n * tiles_per_round / dt)
for k, v in res.items():
    print(f"{k}: {np.median(v):.1f} tiles/s (min {min(v):.1f}, max {max(v):.1f})")
