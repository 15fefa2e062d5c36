"""The MLP's row parts on batches that are not a multiple of 16 sub-tiles (round 6): the reference's default geometry (1024-px tiles: 25 sub-tiles, 8 tiles
per launch = 200) and TTA / 512-px batches of 144.  Network alone (cpx_net_forward), debug build, parts on (production) against off, interleaved."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, engine, synth
dev = torch.device("cuda:0"); L = _lib.lib()
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
for nS in (200, 144, 56, 32):
    patches = (torch.randn(nS * 1024, 192, device=dev) * 0.5).to(torch.bfloat16)
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=dev)
    ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def fwd(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(), ws.numel(), st))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    res = {0: [], 1: []}
    fwd(2)
    for rnd in range(4):
        for v in ((1, 0) if rnd % 2 == 0 else (0, 1)):
            L.cpx_net_set_mlp_parts(v); res[v].append(fwd(3))
    L.cpx_net_set_mlp_parts(1)
    a, b = float(np.median(res[0])), float(np.median(res[1]))
    print(f"{nS:4d} sub-tiles: unsplit {a:8.2f} ms ({nS / a * 1e3:7.1f} sub-tiles/s)   {L.cpx_net_mlp_parts(nS, 0)} row parts {b:8.2f} ms ({nS / b * 1e3:7.1f} sub-tiles/s)   parts / unsplit {b / a:.4f}", flush=True)
