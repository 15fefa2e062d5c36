"""Network-only step time (cpx_net_forward back to back) vs the full engine step, one process."""
import sys, os, time, ctypes as C
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
    eng.result(prev)
sl = eng.slots[0]
def net(n):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        _lib.check(L.cpx_net_forward(C.byref(w.c), sl.patches.data_ptr(), 32, sl.head.data_ptr(), eng.net_ws.data_ptr(),
                                     eng.net_ws_bytes, st))
steps(3); net(2); torch.cuda.synchronize()
res = {"engine": [], "net": []}
for rnd in range(4):
    t = time.perf_counter(); steps(20); torch.cuda.synchronize(); res["engine"].append((time.perf_counter() - t) / 20 * 1e3)
    t = time.perf_counter(); net(20); torch.cuda.synchronize(); res["net"].append((time.perf_counter() - t) / 20 * 1e3)
for k, v in res.items(): print(k, "ms/step: min %.3f median %.3f" % (min(v), sorted(v)[2]), [round(x, 2) for x in v])
