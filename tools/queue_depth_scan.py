"""Does the depth of the launch queue change what the device needs for a step?  cpx_net_forward back to back (175 launches each), the host
kept at most `ahead` forwards in front of the device (event synchronise), per-forward device time from events; then the whole engine step
(network + post-processing on its own stream) the same way.  Round 5: bench.py's loop queued ~12 steps ahead and its steps took 21.8 ms;
paced one step ahead like the CLI's tile loop, 21.2 ms."""
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
sl = eng.slots[0]
def net_once():
    _lib.check(L.cpx_net_forward(C.byref(w.c), sl.patches.data_ptr(), 32, sl.head.data_ptr(), eng.net_ws.data_ptr(), eng.net_ws_bytes,
                                 torch.cuda.current_stream().cuda_stream))
def run(kind, n, ahead):
    evs, prev = [torch.cuda.Event(enable_timing=True)], None
    evs[0].record()
    t0 = time.perf_counter()
    for i in range(n):
        if kind == "net":
            net_once()
        else:
            sid = eng.submit(tiles, inject=inj, records=True)
            if prev is not None: eng.result(prev)
            prev = sid
        evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record(eng.s_net if kind == "engine" else None)
        if ahead and i >= ahead: evs[i + 1 - ahead].synchronize()
    if prev is not None: eng.result(prev)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    d = sorted(a.elapsed_time(b) for a, b in zip(evs[3:], evs[4:]))
    return wall, d[len(d) // 2], d[0], d[-1]
for kind in ("net", "engine"):
    run(kind, 5, 1)
    for ahead in (1, 2, 4, 8, 0):
        for rep in range(2):
            wall, med, lo, hi = run(kind, 40, ahead)
            print(f"{kind:6s} host at most {ahead if ahead else 'unbounded':>9} ahead: wall {wall:7.3f} ms per step, device per step median {med:7.3f} (min {lo:.3f}, max {hi:.3f})")
