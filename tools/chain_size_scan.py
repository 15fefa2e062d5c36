"""The fused post-processing chain (cpx_compute_masks) by instance size: 8 tiles of 256 x 256 with a grid of discs of radius R whose flows point at
their centres (|dP| = 5 x unit vector, cellprob > 0 inside).  A data-dependent slow path anywhere in the chain shows as a step in the curve
(round 5 found one in the diffusion stage: tools/diffuse_size_scan.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import ops
dev = torch.device("cuda:0")
H = W = int(os.environ.get("CHAIN_HW", "256")); NT = int(os.environ.get("CHAIN_NT", "8"))     # (CHAIN_HW=1024 CHAIN_NT=2: the CLI's default tile)
yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
RADII = [int(r) for r in sys.argv[1:]] or [2, 3, 5, 8, 12, 16, 20, 24, 26, 28, 32, 40, 48, 60]      # (one radius on the command line: for rocprofv3 --kernel-trace --stats)
for R in RADII:
    dP = np.zeros((2, H, W), np.float32); cp = np.full((H, W), -5.0, np.float32)
    step = 2 * R + 5
    n = 0
    for cy in range(R + 2, H - R - 2, step):
        for cx in range(R + 2, W - R - 2, step):
            dy, dx = cy - yy, cx - xx
            r = np.sqrt(dy * dy + dx * dx)
            inside = r <= R
            k = 5.0 / np.maximum(r, 1.0)
            dP[0][inside] = (dy * k)[inside]; dP[1][inside] = (dx * k)[inside]
            cp[inside] = 5.0
            n += 1
    dPd = torch.from_numpy(np.stack([dP] * NT)).to(dev); cpd = torch.from_numpy(np.stack([cp] * NT)).to(dev)
    lg = torch.zeros((NT, 7, H, W), dtype=torch.float32, device=dev); lg[:, 1] = 1.0
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m, cm, nl = ops.compute_masks(dPd, cpd, lg); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"radius {R:3d} (box {2 * R + 1:3d}, padded {(2 * R + 3) ** 2:5d} cells): {n:3d} discs per tile, {int(nl[0]):3d} instances found, foreground {float((cpd > 0).float().mean()):.2f}: "
          f"{sorted(ts)[2]:8.1f} us per batch of {NT} tiles of {H} x {W}")
