"""Where do the ~1 ms launches come from?  (Round-5 review: BENCH_r05 recorded ONE attention launch of 1.013 ms among 144 of 0.183 ms; the round-6
baseline run has one mlp.lin1 launch of 1.036 ms.)

Every launch of the five per-layer kernels gets its own HIP-event pair (cpx_prof_create, stride 1) over N steps, in three modes:
  engine   the bench's loop: Engine.submit (network stream + post-processing stream, flow injection) and a per-step read-back
  net      cpx_net_forward alone on one stream (no post-processing, no copies), one host sync per step
  net+spin ... with the host thread delayed by a 2 ms busy wait in front of every 7th step's launches (does a late host show up as a long launch?)
  engine+pin / +malloc / +free   the engine loop with, while the device is busy with steps 20 and 40, a fresh 64 MB pinned host allocation
           (hipHostMalloc: what a new TileStream's pin_memory() does) / a fresh 256 MB device allocation (hipMalloc) / torch.cuda.empty_cache() (hipFree)
Prints, per mode, the launches longer than 2 x the median of their kind with (step, layer, kind), and the per-step host issue time."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from classpose_amd import _lib, engine, synth

dev = torch.device("cuda:0")
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
depth, bt, T, ncls = 24, 8, 256, 7
sd = synth.make_state_dict(ncls, None, depth=depth, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
eng = engine.Engine(w, T, batch_tiles=bt)
tiles = torch.from_numpy(np.stack([synth.render_region(1234, 224 * i, 448, T, T) for i in range(bt)])).to(dev)
f = [synth.analytic_fields(1234, 224 * i, 448, T, T, ncls) for i in range(bt)]
inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
parts = int(L.cpx_net_mlp_parts(bt * eng.n_sub, _lib.DTYPE_CODE["bf16"]))
per_fwd = depth * (3 + 2 * parts)
KINDS = _lib.PROF_KINDS


def report(mode, host_ms):
    cap = N * per_fwd + 8
    ms, kind, n = (C.c_float * cap)(), (C.c_int * cap)(), C.c_int(0)
    _lib.check(L.cpx_prof_collect_launches(prof, ms, kind, cap, C.byref(n)), "collect")
    n = n.value
    assert n == N * per_fwd, (n, N * per_fwd)
    by = {}
    for i in range(n):
        by.setdefault(kind[i], []).append(ms[i])
    med = {k: float(np.median(v)) for k, v in by.items()}
    print(f"== {mode}: {N} steps, {n} timed launches; medians (us): " + ", ".join(f"{KINDS[k]} {med[k] * 1e3:.1f}" for k in sorted(med)))
    out = [(i // per_fwd, i % per_fwd, kind[i], ms[i]) for i in range(n) if ms[i] > 2.0 * med[kind[i]]]
    for step, pos, k, v in out:
        print(f"   step {step:3d}  launch {pos:3d} of the forward (layer {pos // (3 + 2 * parts)})  {KINDS[k]:9s} {v * 1e3:8.1f} us  = median + {(v - med[k]) * 1e3:.0f} us")
    if not out:
        print("   no launch above 2 x its kind's median")
    print("   host ms per step (issue + wait): min %.2f median %.2f max %.2f" % (min(host_ms), float(np.median(host_ms)), max(host_ms)), flush=True)


for mode in ("engine", "net", "net+spin", "engine+pin", "engine+malloc", "engine+free"):
    prof = C.c_void_p()
    _lib.check(L.cpx_prof_create(N * per_fwd + 8, 1, 0x1F, C.byref(prof)), "prof_create")
    host = []
    if mode.startswith("engine"):
        for _ in range(3):
            eng.result(eng.submit(tiles, inject=inj, records=True))
        torch.cuda.synchronize()
        eng.w.c.prof = prof
        prev = None
        for s in range(N):
            t = time.perf_counter()
            sid = eng.submit(tiles, inject=inj, records=True)
            if s in (20, 40):
                t_x = time.perf_counter()
                if mode == "engine+pin":
                    keep = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
                elif mode == "engine+malloc":
                    keep2 = torch.empty((256 << 20) + s * 4096, dtype=torch.uint8, device=dev)
                elif mode == "engine+free":
                    keep2 = None
                    torch.cuda.empty_cache()
                if mode != "engine":
                    print(f"   step {s}: the host call took {(time.perf_counter() - t_x) * 1e3:.2f} ms")
            if prev is not None:
                out = eng.result(prev)
                int(out.nlabels.max())                     # the read-back the CLI's loop waits for
            prev = sid
            host.append((time.perf_counter() - t) * 1e3)
        eng.result(prev)
        torch.cuda.synchronize()
        eng.w.c.prof = None
    else:
        sl = eng.slots[0]
        st = torch.cuda.Stream(device=dev)
        def fwd():
            _lib.check(L.cpx_net_forward(C.byref(eng.w.c), sl.patches.data_ptr(), bt * eng.n_sub, sl.head.data_ptr(), eng.net_ws.data_ptr(),
                                         eng.net_ws_bytes, st.cuda_stream), "net_forward")
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        eng.w.c.prof = prof
        for s in range(N):
            t = time.perf_counter()
            if mode == "net+spin" and s % 7 == 3:
                t_end = time.perf_counter() + 2e-3
                while time.perf_counter() < t_end:
                    pass
            fwd()
            st.synchronize()
            host.append((time.perf_counter() - t) * 1e3)
        eng.w.c.prof = None
    report(mode, host)
    L.cpx_prof_destroy(prof)
