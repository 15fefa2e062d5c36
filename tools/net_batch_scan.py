"""Network time per sub-tile against the number of sub-tiles per launch (cpx_net_forward back to back): does a working set
that fits the 256 MiB Infinity Cache (16 sub-tiles: hidden tensor 134 MB) beat the BASELINE batch of 32 (268 MB)?"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, engine, synth
dev = torch.device("cuda:0"); L = _lib.lib()
sd = synth.make_state_dict(7, None, depth=24, seed=0)
w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
for nS in ([int(a) for a in sys.argv[1:]] or (8, 16, 24, 32, 48, 64, 96)):
    nbytes = L.cpx_net_workspace_bytes(nS, w.c.dtype)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    patches = (torch.randn(nS * 1024, 192, device=dev) * 0.5).to(torch.bfloat16)
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def net(n):
        for _ in range(n):
            _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(), nbytes, st))
    net(3); torch.cuda.synchronize()
    ts = []
    for rnd in range(4):
        t = time.perf_counter(); net(10); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 10 * 1e3)
    m = sorted(ts)[1]
    print(f"nS={nS:3d}: {m:8.3f} ms per forward  {m / nS * 1e3:7.1f} us per sub-tile  {nS * 727.3e9 / m / 1e9:7.1f} TFLOP/s", flush=True)
    del ws, patches, head
