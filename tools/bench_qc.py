"""Timing of the GrandQC network (cpx_qc_forward) on 512x512 patches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import grandqc, synth
dev = torch.device("cuda:0")
for ncls, nB in ((2, 1), (2, 8), (8, 8)):
    net = grandqc.QcNet.from_state_dict(synth.make_grandqc_state_dict(ncls, 0), dev)
    x = torch.from_numpy(np.stack([synth.render_region(1, 600 * i, 0, 512, 512) for i in range(nB)])).to(dev)
    for _ in range(2): net.forward(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): net.forward(x)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    print(f"classes {ncls} batch {nB}: {dt*1e3:.1f} ms / forward = {dt/nB*1e3:.1f} ms / patch  (~{45.0*nB/dt/1e3:.1f} TFLOP/s f32)", flush=True)
