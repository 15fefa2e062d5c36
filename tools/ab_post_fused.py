"""Post-processing chain of one 8-tile batch (ids + classes + per-cell records), one process, interleaved: the fused chain of cpx_compute_masks_records against the stage-wise sequence it replaces (cpx_postproc_set_fused(0): 38 launches).
Device time by HIP events over back-to-back repetitions, host issue time by the host clock, launches counted by the library."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, synth
dev = torch.device("cuda:0"); L = _lib.lib()
nT, H, W = 8, 256, 256
f = [synth.analytic_fields(1234, 224 * i, 0, 256, 256, 7) for i in range(nT)]
dP, cp, lg = (torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3))
max_rec = min(L.cpx_postproc_max_labels(H, W), 65535)
masks = torch.zeros((nT, H, W), dtype=torch.int16, device=dev); cm = torch.zeros((nT, H, W), dtype=torch.uint8, device=dev)
nlab = torch.zeros(nT, dtype=torch.int32, device=dev); cnt = torch.zeros(nT, dtype=torch.int32, device=dev)
rec = torch.zeros(nT * max_rec * C.sizeof(_lib.CpxRecord), dtype=torch.uint8, device=dev)
ws = torch.empty(L.cpx_postproc_workspace_bytes(nT, H, W), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def once():
    _lib.check(L.cpx_compute_masks_records(dP.data_ptr(), cp.data_ptr(), lg.data_ptr(), nT, 7, H, W, 0.0, 0.4, 200, 15, 0.4, masks.data_ptr(),
                                           cm.data_ptr(), nlab.data_ptr(), max_rec, rec.data_ptr(), cnt.data_ptr(), ws.data_ptr(), st))
res = {0: [], 1: []}; host = {0: [], 1: []}; launches = {}; outs = {}
for rnd in range(6):
    for fused in (1, 0):
        L.cpx_postproc_set_fused(fused)
        n0 = L.cpx_postproc_launch_count(); once(); launches[fused] = int(L.cpx_postproc_launch_count() - n0)
        for _ in range(3): once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(30): once()
        e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
        res[fused].append(e0.elapsed_time(e1) / 30 * 1e3); host[fused].append((t1 - t0) / 30 * 1e6)
        outs[fused] = (masks.clone(), cm.clone(), nlab.clone(), cnt.clone(), rec.clone())
L.cpx_postproc_set_fused(1)
print("outputs bitwise equal (ids, classes, label counts, record counts, records):", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
for fused in (1, 0):
    print(f"{'fused chain' if fused else 'stage-wise  '}: {launches[fused]:2d} launches  device {np.median(res[fused]):6.1f} us per 8-tile batch (min {min(res[fused]):6.1f})  "
          f"host issue {np.median(host[fused]):6.1f} us   {8 * 2.82e6 / np.median(res[fused]) / 1e3:6.1f} GB/s of the algorithmic 2.82 MB / tile")
