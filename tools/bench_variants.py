"""Engine throughput of the other BASELINE configurations on one GPU (informative; bench.py is configs[1])."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import engine, synth
dev = torch.device("cuda:0")
variants = [("configs[1] conic 256/32 bf16", 7, 256, False, "bf16"),
            ("configs[3] conic 256/32 bf16 --tta", 7, 256, True, "bf16"),
            ("configs[4] 512-px tiles fp16 (semantic head, 7 classes)", 7, 512, False, "fp16"),
            ("configs[2] puma 1024/64 bf16 (10 classes)", 10, 1024, False, "bf16")]
targets = [int(a) for a in sys.argv[1:]] or [32]
for name, ncls, T, tta, prec in variants:
    sd = synth.make_state_dict(ncls, None, depth=24, seed=0)
    w = engine.NetWeights.from_state_dict(sd, prec, dev)
    n_sub = engine.make_tiling(T, T, 256, tta).ny ** 2
    for target in targets:
        nT = max(1, target // n_sub)
        eng = engine.Engine(w, T, batch_tiles=nT, augment=tta)
        tiles = torch.from_numpy(np.stack([synth.render_region(1234, (T - 32) * i, 0, T, T) for i in range(nT)])).to(dev)
        f = [synth.analytic_fields(1234, (T - 32) * i, 0, T, T, ncls) for i in range(nT)]
        inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))

        def steps(n):
            prev = None
            for _ in range(n):
                sid = eng.submit(tiles, inject=inj, records=True)
                if prev is not None:
                    eng.result(prev)
                prev = sid
            eng.result(prev)
        steps(3); torch.cuda.synchronize()
        t = time.perf_counter(); n = 20; steps(n); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{name}: {nT} tiles x {n_sub} sub-tiles per step, {dt/n*1e3:.2f} ms/step, {n*nT/dt:.1f} tiles/s, "
              f"{n*nT*n_sub/dt:.0f} sub-tiles/s, {n*nT*T*T/dt/1e6:.1f} Mpx/s", flush=True)
        del eng
        torch.cuda.empty_cache()
    del w
