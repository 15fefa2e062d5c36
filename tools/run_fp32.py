"""--precision fp32 engine steps (exact-f32 MFMA network) for rocprofv3 --kernel-trace --stats: 3 steps of 8 tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import engine, synth
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "fp32", dev)
eng = engine.Engine(w, 256, batch_tiles=8)
tiles = torch.from_numpy(np.stack([synth.render_region(1234, 224 * i, 0, 256, 256) for i in range(8)])).to(dev)
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
for _ in range(4):
    out = eng.run(tiles, inject=inj)
torch.cuda.synchronize()
print("fp32 steps done,", int(out.nlabels.sum()), "cells in the last batch")
