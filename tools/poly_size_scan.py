"""The engine's polygon stage (cpx_polygonize_device behind the fused chain, as the CLI runs it) by instance size and count: 8 tiles of 256 x 256
filled with discs of radius R (flows injected), one engine step with polygons=...; reports the step's post-stream time (chain + polygons), the
vertices the batch produced against the device pool, and whether fetch_polygons had to give the batch to the host polygoniser (pool overflow)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import engine, synth
dev = torch.device("cuda:0")
H = W = 256; NT = 8
sd = synth.make_state_dict(7, None, depth=1, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
eng = engine.Engine(w, H, batch_tiles=NT)
tiles = torch.zeros((NT, H, W, 3), dtype=torch.uint8, device=dev)
yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
print("device vertex pool:", eng.max_pts, "vertices for", NT, "tiles")
for R in [int(r) for r in sys.argv[1:]] or [2, 3, 4, 5, 8, 12, 20, 32, 60]:
    dP = np.zeros((2, H, W), np.float32); cp = np.full((H, W), -5.0, np.float32)
    step = 2 * R + 3
    n = 0
    for cy in range(R + 1, H - R - 1, step):
        for cx in range(R + 1, W - R - 1, step):
            dy, dx = cy - yy, cx - xx
            r = np.sqrt(dy * dy + dx * dx)
            inside = r <= R
            k = 5.0 / np.maximum(r, 1.0)
            dP[0][inside] = (dy * k)[inside]; dP[1][inside] = (dx * k)[inside]
            cp[inside] = 5.0
            n += 1
    lg = np.zeros((7, H, W), np.float32); lg[1] = 1.0
    inj = tuple(torch.from_numpy(np.stack([a] * NT)).to(dev) for a in (dP, cp, lg))
    origins = [(300 * i, 0) for i in range(NT)]
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.run(tiles, inject=inj, records=True, polygons=(1.0, origins))
        polys = eng.fetch_polygons(NT, out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    total = int(out.n_pts_total.item())
    print(f"radius {R:3d}: {n:4d} discs per tile, {int(out.nlabels[0]):4d} instances, {total:7d} vertices in the batch ({total / max(int(out.nlabels.sum()), 1):5.1f} per cell) "
          f"-> {'DEVICE polygons' if polys is not None else 'POOL OVERFLOW: host polygoniser'}; engine step incl. read-back {sorted(ts)[1]:7.2f} ms")
