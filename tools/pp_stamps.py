"""Where do the two workgroups of a CU stand relative to each other in the ping-pong GEMM?  s_memtime stamps (debug build,
cpx_gemm_set_dbg(32)) at tile start / main-loop end / tile end of every workgroup: per-phase durations, and the share of
each epilogue that runs beside its CU partner's main loop."""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collections import defaultdict
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
g = torch.Generator().manual_seed(0)
M, N, K = 32768, 4096, 1024
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
stats = ops.row_stats(A); cs = W.float().sum(1).contiguous()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["gelu"], b.data_ptr(), None, out.data_ptr(), N,
                             stats.data_ptr(), cs.data_ptr(), None, st))
L.cpx_gemm_set_pingpong(1)
for opts in ((1, 0), (1, 2), (1, 4)):
    L.cpx_gemm_set_pingpong_opts(*opts)
    L.cpx_gemm_set_dbg(0)
    for _ in range(5): run()
    L.cpx_gemm_set_dbg(32)
    run(); torch.cuda.synchronize()
    L.cpx_gemm_set_dbg(0)
    buf = np.zeros(512 * 98, np.uint64)
    assert L.cpx_gemm_pingpong_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(512, 98)
    key, arr = s[:, 0].astype(np.int64), s[:, 1].astype(np.int64)
    t = s[:, 2:2 + 3 * 8].reshape(512, 8, 3).astype(np.int64)
    main, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1]
    gap = t[:, 1:, 0] - t[:, :-1, 2]
    print(f"options persistent={opts[0]} delay={opts[1]}: distinct CU keys {len(set(key.tolist()))}, workgroups per key "
          f"{np.bincount(np.unique(key, return_counts=True)[1]).tolist()}, arrival parity counts {np.bincount(arr & 1).tolist()}")
    base = t[:, 0, 0].min()
    print(f"   per tile (cycles): main loop median {int(np.median(main))} (p10 {int(np.percentile(main, 10))}, p90 {int(np.percentile(main, 90))}), "
          f"epilogue median {int(np.median(epi))} (p10 {int(np.percentile(epi, 10))}, p90 {int(np.percentile(epi, 90))}), whole launch "
          f"{int(t[:, :, 2].max() - base)}")
    by = defaultdict(list)
    for w in range(512):
        by[int(key[w])].append(w)
    ov = []
    for k_, ws in by.items():
        if len(ws) != 2:
            continue
        a_, b_ = ws
        for x, y in ((a_, b_), (b_, a_)):
            for i in range(8):
                e0, e1 = t[x, i, 1], t[x, i, 2]
                inside = 0
                for j in range(8):
                    m0_, m1_ = t[y, j, 0], t[y, j, 1]
                    inside += max(0, min(e1, m1_) - max(e0, m0_))
                ov.append(inside / max(e1 - e0, 1))
    print(f"   share of an epilogue that runs beside the partner's main loop: median {np.median(ov):.2f}, mean {np.mean(ov):.2f}")
    first = np.array([t[ws[1], 0, 0] - t[ws[0], 0, 0] for ws in by.values() if len(ws) == 2])
    print(f"   start offset between the partners' first tiles: median |d| {int(np.median(np.abs(first)))} cycles")
