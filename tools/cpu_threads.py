"""How many torch threads should the CPU baseline use on this host? (1 sub-tile, depth 2)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import synth
from oracle import net
sd = synth.make_state_dict(7, None, depth=2, seed=0)
x = np.random.default_rng(0).random((4, 3, 256, 256)).astype(np.float32)
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    fw = net.make_forward(sd)
    fw(x[:1])
    t = time.perf_counter(); fw(x); dt = time.perf_counter() - t
    print(f"threads {nt:4d}: {dt:.3f} s for 4 sub-tiles x depth 2 -> est {dt*12:.1f} s / tile at depth 24", flush=True)
