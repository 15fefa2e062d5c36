"""Engine step time with the post-processing stream at normal and at high priority (two engines, interleaved rounds, one
process) + the time from a batch's network end to its post-processing end (how long the hidden chain really takes in situ)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import engine, synth
dev = torch.device("cuda:0")
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
engs = {}
for pr in (0, -1):
    os.environ["CPX_POST_STREAM_PRIORITY"] = str(pr)
    engs[pr] = engine.Engine(w, 256, batch_tiles=8)
# round 5: the NETWORK stream at high priority, the post stream normal (key 1)
os.environ["CPX_POST_STREAM_PRIORITY"] = "0"; os.environ["CPX_NET_STREAM_PRIORITY"] = "-1"
engs[1] = engine.Engine(w, 256, batch_tiles=8)
os.environ["CPX_NET_STREAM_PRIORITY"] = "0"
tiles = torch.from_numpy(np.stack([synth.render_region(1234, 224 * i, 0, 256, 256) for i in range(8)])).to(dev)
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(8)]
inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
def steps(eng, n):
    prev = None
    for _ in range(n):
        sid = eng.submit(tiles, inject=inj, records=True)
        if prev is not None: eng.result(prev)
        prev = sid
    return eng.result(prev)
res = {0: [], -1: [], 1: []}; outs = {}
for pr in (0, -1, 1): steps(engs[pr], 3)
for rnd in range(5):
    for pr in (0, -1, 1):
        steps(engs[pr], 2); torch.cuda.synchronize(); t = time.perf_counter(); o = steps(engs[pr], 20); torch.cuda.synchronize()
        res[pr].append((time.perf_counter() - t) / 20 * 1e3); outs[pr] = o.masks.clone()
print("identical outputs:", torch.equal(outs[0], outs[-1]))
for pr in (0, -1, 1):
    print(f"{'network stream HIGH, post normal' if pr == 1 else 'post stream priority %2d' % pr}: engine ms/step min {min(res[pr]):.3f} median {sorted(res[pr])[2]:.3f}  {[round(x, 2) for x in res[pr]]}")
