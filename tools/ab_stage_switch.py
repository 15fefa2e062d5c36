"""A/B of a debug-build switch with the per-stage view: interleaved in one process, the whole engine step, the network alone, and the
launch medians of its stages (HIP events per launch, cpx_prof_*); engine outputs compared bit for bit.
usage: python tools/ab_stage_switch.py cpx_gemm_set_nt=0,1 [restore_value [other_setter=value ...]]   (the others are set once, before the A/B)"""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, engine, synth
name, vs = sys.argv[1].split("=")
vals = tuple(int(x) for x in vs.split(","))
restore = int(sys.argv[2]) if len(sys.argv) > 2 else vals[0]
dev = torch.device("cuda:0"); L = _lib.lib()
setter = getattr(L, name)
for fixed in sys.argv[3:]:
    fn, fv = fixed.split("="); getattr(L, fn)(int(fv)); print("fixed:", fn, fv)
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
    return eng.result(prev)
sl = eng.slots[0]
def net(n, prof=None):
    st = torch.cuda.current_stream().cuda_stream
    w.c.prof = prof
    for _ in range(n):
        _lib.check(L.cpx_net_forward(C.byref(w.c), sl.patches.data_ptr(), 32, sl.head.data_ptr(), eng.net_ws.data_ptr(), eng.net_ws_bytes, st))
    w.c.prof = None
def stage_medians(n):
    cap = n * (24 * 7 + 2) + 8
    prof = C.c_void_p()
    _lib.check(L.cpx_prof_create(cap, 1, 0x7F, C.byref(prof)))
    net(n, prof); torch.cuda.synchronize()
    ms, kind, cnt = (C.c_float * cap)(), (C.c_int * cap)(), C.c_int(0)
    _lib.check(L.cpx_prof_collect_launches(prof, ms, kind, cap, C.byref(cnt)))
    L.cpx_prof_destroy(prof)
    per = {}
    for i in range(min(cnt.value, cap)): per.setdefault(_lib.PROF_KINDS[kind[i]], []).append(ms[i] * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in per.items()}
steps(3); net(2); torch.cuda.synchronize()
res = {v: {"engine": [], "net": [], "stages": []} for v in vals}
outs = {}
for rnd in range(5):
    for v in vals:
        setter(v)
        steps(2); torch.cuda.synchronize()
        t = time.perf_counter(); o = steps(20); torch.cuda.synchronize(); res[v]["engine"].append((time.perf_counter() - t) / 20 * 1e3)
        outs[v] = (o.dP.clone(), o.masks.clone())
        net(2); torch.cuda.synchronize()
        t = time.perf_counter(); net(20); torch.cuda.synchronize(); res[v]["net"].append((time.perf_counter() - t) / 20 * 1e3)
        res[v]["stages"].append(stage_medians(3))
setter(restore)
print(name, "bitwise equal engine outputs:", all(torch.equal(outs[vals[0]][i], outs[v][i]) for v in vals for i in (0, 1)))
med = lambda v: sorted(v)[len(v) // 2]
for v in vals:
    r = res[v]
    print("%s(%d): engine step median %.3f ms (min %.3f)  network alone median %.3f ms (min %.3f)" %
          (name, v, med(r["engine"]), min(r["engine"]), med(r["net"]), min(r["net"])))
    print("         launch medians [us]:", {k: round(med([s[k] for s in r["stages"]]), 1) for k in r["stages"][0]})
