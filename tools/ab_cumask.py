"""Post-processing stream restricted to a CU subset (hipExtStreamCreateWithCUMask): engine step A/B."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")      # the A/B switches live in the -DCPX_DEBUG library (libclasspose_hip_debug.so)
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, engine, synth
dev = torch.device("cuda:0"); L = _lib.lib()
torch.cuda.set_device(dev); torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    """bits: list of CU indices (bit positions) enabled"""
    words = [0] * 8
    for b in bits: words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)
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
normal = eng.s_post
variants = {"all CUs": normal}
for name, bits in (("32 CUs (bits 0-31)", range(32)), ("64 CUs (bits 0-63)", range(64)), ("16 CUs (bits 0-15)", range(16)),
                   ("32 CUs (every 8th bit)", range(0, 256, 8))):
    variants[name] = masked_stream(list(bits))
ref = steps(3); ref_m = ref.masks.clone(); torch.cuda.synchronize()
res = {k: [] for k in variants}
for rnd in range(4):
    for k, st in variants.items():
        eng.s_post = st
        o = steps(3); torch.cuda.synchronize()
        assert torch.equal(o.masks, ref_m)
        t = time.perf_counter(); steps(20); torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / 20 * 1e3)
for k, v in res.items(): print(f"{k:28s} ms/step: min {min(v):.3f} median {sorted(v)[2]:.3f}")
