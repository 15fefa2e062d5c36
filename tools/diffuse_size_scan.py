"""k_diffuse + k_flow_err_label (cpx_remove_bad_flow_masks) on 8 tiles of 256 x 256 with a 3 x 3 grid of square labels of side s: the cost of a
label by the path its padded box takes -- LDS planes up to 2944 cells (side <= 52), the global planes beyond."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
dP = torch.from_numpy((rng.standard_normal((8, 2, 256, 256)) * 2).astype(np.float32)).to(dev)
for s in (16, 30, 44, 50, 52, 53, 54, 58, 62, 70, 80):
    m = np.zeros((8, 256, 256), np.int32)
    lab = 0
    for gy in range(3):
        for gx in range(3):
            lab += 1
            m[:, 2 + gy * 84: 2 + gy * 84 + s, 2 + gx * 84: 2 + gx * 84 + s] = lab
    masks = torch.from_numpy(m).to(dev)
    ts = []
    for rep in range(6):
        mm = masks.clone(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.remove_bad_flow_masks(mm, dP, 1e9); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"side {s:3d}: padded box {(s + 2) ** 2:5d} cells, niter {2 * (2 * s + 2):4d}: {sorted(ts)[2]:8.1f} us per 8-tile batch (9 labels per tile)")
