"""Old library against new library, one process, interleaved: the five per-layer kernels of the ViT-L block in their
in-engine configuration (qkv = folded LayerNorm + V^T image, attention, proj / mlp.lin2 = residual + LayerNorm statistics,
mlp.lin1 = folded LayerNorm + GELU) at 32 sub-tiles, timed with device events, outputs compared bit for bit.
    python tools/ab_libs.py tools/_ab/libclasspose_hip_r03.so classpose_amd/libclasspose_hip.so [n_subtiles] [rounds]
Both files are PRODUCT libraries (the C ABI of include/classpose_hip.h); ctypes loads each privately (RTLD_LOCAL)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

paths = sys.argv[1:3]
nS = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
vp, ip = C.c_void_p, C.c_int
libs = []
for p in paths:
    L = C.CDLL(os.path.abspath(p))
    L.cpx_gemm_ln.restype = ip
    L.cpx_gemm_ln.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, vp, ip, vp, vp, vp, vp]
    L.cpx_attention.restype = ip
    L.cpx_attention.argtypes = [ip, vp, vp, vp, ip, vp, vp, vp]
    L.cpx_row_stats.restype = ip
    L.cpx_row_stats.argtypes = [vp, ip, vp, vp]
    libs.append(L)
dev = torch.device("cuda:0")
M = nS * 1024
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale)
x = rnd(M, 1024).to(bf).to(dev)
hid = rnd(M, 4096).to(bf).to(dev)
Wqkv = rnd(3072, 1024, scale=1 / 32).to(bf).to(dev); bqkv = rnd(3072).to(dev)
Wproj = rnd(1024, 1024, scale=1 / 32).to(bf).to(dev); bproj = rnd(1024).to(dev)
W1 = rnd(4096, 1024, scale=1 / 32).to(bf).to(dev); b1 = rnd(4096).to(dev)
W2 = rnd(1024, 4096, scale=1 / 64).to(bf).to(dev); b2 = rnd(1024).to(dev)
cs_qkv = Wqkv.float().sum(1).contiguous(); cs1 = W1.float().sum(1).contiguous()
rel_h = (rnd(64, 64, scale=0.2)).to(bf).to(dev); rel_w = (rnd(64, 64, scale=0.2)).to(bf).to(dev)
rel_h[63] = 0; rel_w[63] = 0
stats = torch.empty((M, 4, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert libs[0].cpx_row_stats(x.data_ptr(), M, stats.data_ptr(), st) == 0
qkv_in = (rnd(M, 3072, scale=1.0)).to(bf).to(dev)
EPI = dict(bf16=0, gelu=1, resid=2, qkv=6)

def outs():
    return dict(qkv=torch.empty((M, 3072), dtype=bf, device=dev), vT=torch.empty((M, 1024), dtype=bf, device=dev),
                attn=torch.empty((M, 1024), dtype=bf, device=dev), avT=torch.empty((M, 1024), dtype=bf, device=dev),
                proj=torch.empty((M, 1024), dtype=bf, device=dev), pst=torch.zeros((M, 4, 2), dtype=torch.float32, device=dev),
                fc1=torch.empty((M, 4096), dtype=bf, device=dev),
                fc2=torch.empty((M, 1024), dtype=bf, device=dev), fst=torch.zeros((M, 4, 2), dtype=torch.float32, device=dev))
O = [outs(), outs()]
def ck(rc):
    assert rc == 0, rc
OPS = {
    "qkv (LN + V^T)": lambda L, o: ck(L.cpx_gemm_ln(x.data_ptr(), Wqkv.data_ptr(), M, 3072, 1024, EPI["qkv"], bqkv.data_ptr(), o["vT"].data_ptr(),
                                                   o["qkv"].data_ptr(), 3072, stats.data_ptr(), cs_qkv.data_ptr(), None, st)),
    "attention": lambda L, o: ck(L.cpx_attention(0, qkv_in.data_ptr(), rel_h.data_ptr(), rel_w.data_ptr(), nS, o["avT"].data_ptr(), o["attn"].data_ptr(), st)),
    "proj (resid + stats)": lambda L, o: ck(L.cpx_gemm_ln(x.data_ptr(), Wproj.data_ptr(), M, 1024, 1024, EPI["resid"], bproj.data_ptr(), x.data_ptr(),
                                                         o["proj"].data_ptr(), 1024, None, None, o["pst"].data_ptr(), st)),
    "mlp.lin1 (LN + GELU)": lambda L, o: ck(L.cpx_gemm_ln(x.data_ptr(), W1.data_ptr(), M, 4096, 1024, EPI["gelu"], b1.data_ptr(), None,
                                                         o["fc1"].data_ptr(), 4096, stats.data_ptr(), cs1.data_ptr(), None, st)),
    "mlp.lin2 (resid + stats)": lambda L, o: ck(L.cpx_gemm_ln(hid.data_ptr(), W2.data_ptr(), M, 1024, 4096, EPI["resid"], b2.data_ptr(), x.data_ptr(),
                                                             o["fc2"].data_ptr(), 1024, None, None, o["fst"].data_ptr(), st)),
}
FL = {"qkv (LN + V^T)": 2.0 * M * 3072 * 1024, "attention": 4.43e9 * nS, "proj (resid + stats)": 2.0 * M * 1024 * 1024,
      "mlp.lin1 (LN + GELU)": 2.0 * M * 4096 * 1024, "mlp.lin2 (resid + stats)": 2.0 * M * 1024 * 4096}
t = {k: ([], []) for k in OPS}
for r in range(rounds):
    for k, f in OPS.items():
        for i, L in enumerate(libs):
            for _ in range(3): f(L, O[i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f(L, O[i])
            e1.record(); torch.cuda.synchronize()
            t[k][i].append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"A = {paths[0]}\nB = {paths[1]}\n{nS} sub-tiles, {rounds} interleaved rounds of 20 launches (us per launch: median / min)")
tot = [0.0, 0.0]
for k, (a, b) in t.items():
    ma, mb = np.median(a), np.median(b)
    tot[0] += ma; tot[1] += mb
    print(f"{k:26s} A {ma:7.1f} / {min(a):7.1f}   B {mb:7.1f} / {min(b):7.1f}   B/A {mb / ma:6.3f}   B: {FL[k] / mb / 1e6:7.1f} TFLOP/s = {FL[k] / mb / 1e6 / 2500:5.3f} of 2.5 PF")
print(f"{'layer sum':26s} A {tot[0]:7.1f}             B {tot[1]:7.1f}             B/A {tot[1] / tot[0]:6.3f}")
same = {k: bool(torch.equal(O[0][k].view(torch.int16) if O[0][k].dtype == bf else O[0][k], O[1][k].view(torch.int16) if O[1][k].dtype == bf else O[1][k])) for k in O[0]}
print("outputs bitwise equal:", same)
if not all(same.values()):
    for k in O[0]:
        if not same[k]:
            d = (O[0][k].float() - O[1][k].float()).abs()
            print(f"  {k}: {int((d > 0).sum())} of {d.numel()} differ, max |diff| {float(d.max()):.3e}, max |A| {float(O[0][k].float().abs().max()):.3e}")
