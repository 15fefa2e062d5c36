"""Persistent 256^2 GEMM epilogue: LDS-staged rows (two halves, four barriers) against DIRECT stores from the accumulator registers
(v_permlane16_swap -> 16-byte buffer stores, one barrier; cpx_gemm_set_direct, debug build).  The four layer shapes in their in-engine
configuration, one process, interleaved in alternating order, outputs compared bit for bit.
    python tools/ab_gemm_direct.py [switch-name [other_switch=value ...]]        (default: direct)"""
import os as _os
_os.environ.setdefault("CLASSPOSE_HIP_DEBUG", "1")
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M = 32768
g = torch.Generator().manual_seed(0); bf = torch.bfloat16
rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
x = rnd(M, 1024).to(bf).to(dev); hid = rnd(M, 4096).to(bf).to(dev)
Wqkv = rnd(3072, 1024, scale=1 / 32).to(bf).to(dev); bqkv = rnd(3072).to(dev)
Wproj = rnd(1024, 1024, scale=1 / 32).to(bf).to(dev); bproj = rnd(1024).to(dev)
W1 = rnd(4096, 1024, scale=1 / 32).to(bf).to(dev); b1 = rnd(4096).to(dev)
W2 = rnd(1024, 4096, scale=1 / 64).to(bf).to(dev); b2 = rnd(1024).to(dev)
cs_qkv = Wqkv.float().sum(1).contiguous(); cs1 = W1.float().sum(1).contiguous()
stats = ops.row_stats(x)
st = torch.cuda.current_stream().cuda_stream
E = ops.EPI
def mk(): return dict(qkv=torch.empty((M, 3072), dtype=bf, device=dev), vT=torch.empty((M, 1024), dtype=bf, device=dev), proj=torch.empty((M, 1024), dtype=bf, device=dev),
                      pst=torch.zeros((M, 4, 2), device=dev), fc1=torch.empty((M, 4096), dtype=bf, device=dev), fc2=torch.empty((M, 1024), dtype=bf, device=dev), fst=torch.zeros((M, 4, 2), device=dev))
O = {0: mk(), 1: mk()}
ck = lambda rc: _lib.check(rc)
OPS = {
    "qkv": lambda o: ck(L.cpx_gemm_ln(x.data_ptr(), Wqkv.data_ptr(), M, 3072, 1024, E["qkv"], bqkv.data_ptr(), o["vT"].data_ptr(), o["qkv"].data_ptr(), 3072, stats.data_ptr(), cs_qkv.data_ptr(), None, st)),
    "proj": lambda o: ck(L.cpx_gemm_ln(x.data_ptr(), Wproj.data_ptr(), M, 1024, 1024, E["resid"], bproj.data_ptr(), x.data_ptr(), o["proj"].data_ptr(), 1024, None, None, o["pst"].data_ptr(), st)),
    "mlp.lin1": lambda o: ck(L.cpx_gemm_ln(x.data_ptr(), W1.data_ptr(), M, 4096, 1024, E["gelu"], b1.data_ptr(), None, o["fc1"].data_ptr(), 4096, stats.data_ptr(), cs1.data_ptr(), None, st)),
    "mlp.lin2": lambda o: ck(L.cpx_gemm_ln(hid.data_ptr(), W2.data_ptr(), M, 1024, 4096, E["resid"], b2.data_ptr(), x.data_ptr(), o["fc2"].data_ptr(), 1024, None, None, o["fst"].data_ptr(), st)),
}
t = {k: {0: [], 1: []} for k in OPS}
SW_NAME = sys.argv[1] if len(sys.argv) > 1 else 'direct'
SW = getattr(L, 'cpx_gemm_set_' + SW_NAME)
for kv in sys.argv[2:]:                      # other switches held fixed, e.g. `balanced direct=0`
    getattr(L, 'cpx_gemm_set_' + kv.split('=')[0])(int(kv.split('=')[1]))
for r in range(8):
    for k, f in OPS.items():
        for v in ((0, 1) if r % 2 == 0 else (1, 0)):
            SW(2 * v if SW_NAME == 'direct' else v)     # direct: 2 = every non-residual epilogue (1, the production setting, is the GELU epilogue only)
            for _ in range(3): f(O[v])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f(O[v])
            e1.record(); torch.cuda.synchronize()
            t[k][v].append(e0.elapsed_time(e1) / 20 * 1e3)
SW(1)
for k in OPS:
    a, b = np.median(t[k][0]), np.median(t[k][1])
    print(f"{k:10s} off {a:7.1f} us (min {min(t[k][0]):7.1f})   on  {b:7.1f} us (min {min(t[k][1]):7.1f})   ratio {b / a:6.3f}")
print("outputs bitwise equal:", {k: bool(torch.equal(O[0][k].view(torch.int16) if O[0][k].dtype == bf else O[0][k], O[1][k].view(torch.int16) if O[1][k].dtype == bf else O[1][k])) for k in O[0]})
