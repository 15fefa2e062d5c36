"""Interleaved A/B of a library switch on the whole engine step (one process, one device)."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, engine, synth
switch = sys.argv[1] if len(sys.argv) > 1 else "cpx_gemm_set_l2_block"
dev = torch.device("cuda:0")
L = _lib.lib()
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
eng = engine.Engine(w, 256, batch_tiles=8)
tiles = torch.from_numpy(np.stack([synth.render_region(1234, 224 * i, 0, 256, 256) for i in range(8)])).to(dev)
def run(n):
    prev = None
    for i in range(n):
        sid = eng.submit(tiles)
        if prev is not None: eng.result(prev)
        prev = sid
    eng.result(prev)
    torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(4):
    for v in (1, 0):
        getattr(L, switch)(v)
        run(2)
        t = time.perf_counter(); run(10); dt = (time.perf_counter() - t) / 10
        res[v].append(dt * 1e3)
getattr(L, switch)(1)
for v in (1, 0):
    print(f"{switch}={v}: ms/step min {min(res[v]):.2f} median {sorted(res[v])[len(res[v])//2]:.2f}  all {[round(x,2) for x in res[v]]}")
