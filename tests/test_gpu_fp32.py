"""--precision fp32 (what every reference integration test passes,
/root/reference/tests/test_prediction_integration.py:67,113,159,199): exact-f32 MFMA kernels vs
float64 torch references of the same op, and the whole ClassTransformer vs the fp32 oracle at the
tightest tolerance of SURVEY 8c item 1: max-abs <= 1e-3 and rel-L2 <= 1e-4."""
import ctypes as C

import numpy as np
import pytest
import torch

from classpose_amd import _lib, engine, ops, synth
from oracle import net as onet
from oracle import tiling

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (256, 384, 192), (1024, 1024, 1024), (2048, 640, 256),
                                   (1024, 256, 2304), (4096, 4096, 1024)])
def test_gemm_f32_epilogues(cuda, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(cuda)
    ref = (A.double() @ W.double().T)
    tol = 2e-6                                  # f32 products / sums, only the summation order differs
    assert _rel(ops.gemm(A, W, "f32", None), ref) < tol
    assert _rel(ops.gemm(A, W, "f32", bias), ref + bias) < tol
    assert _rel(ops.gemm(A, W, "gelu", bias), torch.nn.functional.gelu(ref + bias)) < tol
    assert _rel(ops.gemm(A, W, "relu", bias), torch.relu(ref + bias)) < tol
    assert _rel(ops.gemm(A, W, "resid", bias, res), ref + bias + res) < tol
    if N == 1024:
        pos = torch.randn(1024, N, generator=g).to(cuda)
        exp = ref + bias + pos[torch.arange(M, device=cuda) % 1024]
        assert _rel(ops.gemm(A, W, "pos", bias, pos), exp) < tol


def test_gemm_f32_identity_asymmetric(cuda):
    """A = I against an asymmetric W: transposed / permuted fragment maps show up exactly."""
    A = torch.eye(128, 128, device=cuda)
    W = ((torch.arange(128 * 128).reshape(128, 128) * 7) % 251 - 125).float().to(cuda)
    assert torch.equal(ops.gemm(A, W, "f32", None), W.T.contiguous())
    A2 = torch.zeros(256, 32, device=cuda)
    A2[torch.arange(256), torch.arange(256) % 32] = 1.0
    W2 = ((torch.arange(128 * 32).reshape(128, 32) * 5) % 127 - 63).float().to(cuda)
    assert torch.equal(ops.gemm(A2, W2, "f32", None), W2.T[torch.arange(256, device=cuda) % 32])


@pytest.mark.parametrize("C_", [1024, 256])
def test_layernorm_f32(cuda, C_):
    g = torch.Generator().manual_seed(C_)
    x = (torch.randn(512, C_, generator=g) * 3 + 1).to(cuda)
    w, b = torch.randn(C_, generator=g).to(cuda), torch.randn(C_, generator=g).to(cuda)
    ref = torch.nn.functional.layer_norm(x.double(), (C_,), w.double(), b.double(), 1e-6)
    out = ops.layernorm(x, w, b, 1e-6)
    assert _rel(out, ref) < 1e-6 and float((out - ref).abs().max()) < 2e-5


def _attention_ref64(qkv, relh63, relw63, nS):
    """flash_forward (vit_sam.py:26-65) in float64."""
    B, L, H = nS, 1024, 16
    q, k, v = qkv.double().reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    idx = (torch.arange(32)[:, None] - torch.arange(32)[None, :] + 31).to(qkv.device)
    Rh, Rw = relh63.double()[idx], relw63.double()[idx]
    qhw = q.reshape(B, H, 32, 32, 64)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", qhw, Rh)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", qhw, Rw)
    bias = (rel_h[..., :, None] + rel_w[..., None, :]).reshape(B, H, L, L)
    att = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias, -1)
    return (att @ v).transpose(1, 2).reshape(B * L, 1024)


def test_attention_f32(cuda):
    g = torch.Generator().manual_seed(5)
    nS = 2
    qkv = torch.randn(nS * 1024, 3072, generator=g).to(cuda)
    relh = (torch.randn(63, 64, generator=g) * 0.3).to(cuda)
    relw = (torch.randn(63, 64, generator=g) * 0.3).to(cuda)
    pad = lambda t: torch.cat([t * 8, torch.zeros(1, 64, device=cuda)]).contiguous()
    out = ops.attention(qkv, pad(relh), pad(relw))
    ref = _attention_ref64(qkv, relh, relw, nS)
    assert _rel(out, ref) < 5e-6, _rel(out, ref)
    assert float((out - ref).abs().max()) < 2e-5


def test_attention_f32_spiked_rows(cuda):
    """one key dominates from tile 21 on: the running-max rescale is exercised"""
    g = torch.Generator().manual_seed(6)
    qkv = torch.randn(1024, 3072, generator=g) * 0.1
    qkv[:, :1024] = 1.0
    qkv[700, 1024:2048] = 30.0
    qkv = qkv.to(cuda)
    z = torch.zeros(64, 64, device=cuda)
    out = ops.attention(qkv, z, z)
    ref = _attention_ref64(qkv, torch.zeros(63, 64, device=cuda), torch.zeros(63, 64, device=cuda), 1)
    assert float((out - ref).abs().max()) < 1e-5
    assert torch.allclose(out[5], qkv[700, 2048:], atol=1e-5)


def _forward(w, x, cuda):
    nS = x.shape[0]
    dt = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}[w.c.dtype]
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5) \
        .reshape(nS * 1024, 192).to(dt).to(cuda)
    L = _lib.lib()
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
    nbytes = L.cpx_net_workspace_bytes(nS, w.c.dtype)
    if w.c.n_unet_ops:
        nbytes += L.cpx_unet_workspace_bytes(w.c.unet_ops, w.c.n_unet_ops, nS, w.c.dtype)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream().cuda_stream))
    ncol = w.c.n_head_cols
    out = head[:, :ncol].reshape(nS, 32, 32, ncol // 64, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(nS, ncol // 64, 256, 256)
    out = out.cpu()
    return torch.cat([out[:, 3:], out[:, :3]], 1)        # oracle channel order: [class logits, dY, dX, cellprob]


@pytest.mark.parametrize("depth,nS,fts", [(24, 2, None), (2, 4, [64, 128])])
def test_net_forward_fp32_vs_oracle(cuda, depth, nS, fts):
    """HIP fp32 vs CPU-restatement fp32, max-abs <= 1e-3 and rel-L2 <= 1e-4"""
    sd = synth.make_state_dict(7, fts, depth=depth, seed=3)
    w = engine.NetWeights.from_state_dict(sd, "fp32", cuda)
    assert w.c.dtype == _lib.DT_F32 and not w.c.fuse_ln
    x = np.random.default_rng(0).random((nS, 3, 256, 256)).astype(np.float32)
    ours = _forward(w, x, cuda)
    ref = onet.class_transformer_forward(sd, torch.from_numpy(x))
    rel, mx = _rel(ours, ref), float((ours - ref).abs().max())
    print(f"fp32 depth {depth}: rel-L2 {rel:.3e}, max-abs {mx:.3e} (|ref| max {float(ref.abs().max()):.3f})")
    assert rel <= 1e-4 and mx <= 1e-3


def test_engine_fp32_end_to_end(cuda):
    """tiles -> ids in fp32: network outputs at the fp32 tolerance, ids bit-exact on the device tensors"""
    from oracle import classmask, dynamics
    sd = synth.make_state_dict(7, None, depth=2, seed=4)
    w = engine.NetWeights.from_state_dict(sd, "fp32", cuda)
    eng = engine.Engine(w, 256, batch_tiles=2)
    tiles = np.stack([synth.render_region(1234, 0, 0, 256, 256), synth.render_region(1234, 224, 448, 256, 256)])
    out = eng.run(torch.from_numpy(tiles).to(cuda))
    fw = onet.make_forward(sd)
    for i in range(2):
        dP, cp, yc = tiling.run_net(fw, tiling.normalize_img(tiles[i:i + 1]), batch_size=8)
        for a, b in ((out.dP[i], dP), (out.cellprob[i], cp), (out.logits[i], yc)):
            assert _rel(a.cpu(), torch.from_numpy(b)) <= 1e-4
            assert float((a.cpu() - torch.from_numpy(b)).abs().max()) <= 1e-3
        ref = dynamics.compute_masks(out.dP[i].cpu().numpy(), out.cellprob[i].cpu().numpy())
        assert np.array_equal(ops.masks_to_numpy(out.masks)[i], ref)
        cm, _ = classmask.compute_class_masks(ref, out.logits[i].cpu().numpy())
        assert np.array_equal(out.class_masks[i].cpu().numpy(), cm.astype(np.uint8))


def test_mixed_precision_engines_from_two_threads(cuda):
    """include/classpose_hip.h: thread-safe for distinct streams + workspaces.  A bf16 and an fp16
    engine driven from two Python threads at once (the reference shares one model between inference
    threads, predict_wsi.py:790-797) give bit-identical results to the serial runs."""
    import threading
    sd = synth.make_state_dict(7, None, depth=4, seed=8)
    engs = {p: engine.Engine(engine.NetWeights.from_state_dict(sd, p, cuda), 256, batch_tiles=2) for p in ("bf16", "fp16")}
    tiles = torch.from_numpy(np.stack([synth.render_region(5, 0, 0, 256, 256), synth.render_region(5, 300, 200, 256, 256)])).to(cuda)
    serial = {}
    for p, e in engs.items():
        o = e.run(tiles)
        torch.cuda.synchronize()
        serial[p] = (o.dP.clone(), o.cellprob.clone(), o.logits.clone(), o.masks.clone())
    results, errs = {p: [] for p in engs}, []

    def work(p):
        try:
            torch.cuda.set_device(cuda)
            with torch.cuda.stream(torch.cuda.Stream(cuda)):
                for _ in range(6):
                    o = engs[p].run(tiles)
                    torch.cuda.current_stream().synchronize()
                    results[p].append((o.dP.clone(), o.cellprob.clone(), o.logits.clone(), o.masks.clone()))
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(p,)) for p in engs]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for p in engs:
        for r in results[p]:
            assert all(torch.equal(a, b) for a, b in zip(r, serial[p])), p
    assert not torch.equal(serial["bf16"][0], serial["fp16"][0])
