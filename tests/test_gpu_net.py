"""GPU numerics: MFMA GEMM / LayerNorm / attention / whole ClassTransformer vs torch fp32 and
the CPU oracle.  Tolerances (stated per test) follow SURVEY 8c: half-precision compute vs an
fp32 reference of the same op on the SAME half-rounded inputs."""
import numpy as np
import pytest
import torch

from classpose_amd import _lib, engine, ops, synth
from oracle import net as onet
from oracle import tiling

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (1024, 1024, 1024), (2048, 640, 256),
                                   (1024, 256, 2304)])
def test_gemm_epilogues(cuda, variant, M, N, K):
    with _lib.use_debug_library() as L:          # the register-staged A/B variant only exists in the -DCPX_DEBUG build
        _gemm_epilogues(L, cuda, variant, M, N, K)


def _gemm_epilogues(L, cuda, variant, M, N, K):
    L.cpx_gemm_set_variant(variant)
    try:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
        bias = torch.randn(N, generator=g).to(cuda)
        ref = A.float() @ W.float().T
        out = ops.gemm(A, W, "f32", bias)
        assert _rel(out, ref + bias) < 1e-5                     # fp32 accumulate of exact products
        out = ops.gemm(A, W, "bf16", None)
        assert torch.equal(out, ref.to(torch.bfloat16)) or _rel(out.float(), ref) < 4e-3
        out = ops.gemm(A, W, "gelu", bias)
        assert _rel(out.float(), torch.nn.functional.gelu(ref + bias)) < 4e-3
        out = ops.gemm(A, W, "relu", bias)
        assert _rel(out.float(), torch.relu(ref + bias)) < 4e-3
        res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(cuda)
        out = ops.gemm(A, W, "resid", bias, res)
        assert _rel(out.float(), ref + bias + res.float()) < 4e-3
        if N == 1024:
            pos = torch.randn(1024, N, generator=g).to(cuda)
            out = ops.gemm(A, W, "pos", bias, pos)
            exp = ref + bias + pos[torch.arange(M, device=cuda) % 1024]
            assert _rel(out.float(), exp) < 4e-3
    finally:
        L.cpx_gemm_set_variant(1)


def test_patch_embedding_on_the_256_tile_kernel(cuda):
    """The patch embedding (M x 1024 x 192 = THREE K tiles, bias + float32 positional table, bf16 out) takes the 256^2 kernel
    since round 3 and emits the first layer's LayerNorm row statistics itself: bitwise equal to the 128^2 kernel (same
    fp32 accumulation order), within bf16 rounding of torch fp32, statistics equal to the row sums of the rounded output."""
    M, N, K = 32768, 1024, 192
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    pos = torch.randn(1024, N, generator=g).to(cuda)
    assert _lib.lib().cpx_gemm_uses_big_tile(M, N, K, ops.EPI["pos"]) == 1
    assert _lib.lib().cpx_gemm_uses_big_tile(M, N, K, ops.EPI["bf16"]) == 0          # odd K-tile counts only for this epilogue
    out, st = ops.gemm_ln(A, W, "pos", bias, pos, want_stats=True)
    plain = ops.gemm(A, W, "pos", bias, pos)
    assert torch.equal(out, plain)
    # race screen of the odd-K-tile tail (LDS-DMA behind counted vmcnt, raw barriers): repeated launches under a concurrent
    # memory stream stay bitwise identical
    noise = torch.empty((8192, 8192), device=cuda)
    side = torch.cuda.Stream(cuda)
    for i in range(16):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                noise.normal_()
        o2, s2 = ops.gemm_ln(A, W, "pos", bias, pos, want_stats=True)
        assert torch.equal(o2, out) and torch.equal(s2, st), i
    side.synchronize()
    with _lib.use_debug_library() as L:
        L.cpx_gemm_set_big(0)
        try:
            small = ops.gemm(A, W, "pos", bias, pos)
        finally:
            L.cpx_gemm_set_big(1)
    assert torch.equal(out, small)
    ref = A.float() @ W.float().T + bias + pos[torch.arange(M, device=cuda) % 1024]
    assert _rel(out.float(), ref) < 4e-3
    o = out.double()
    s1, s2 = st[:, :, 0].double().sum(1), st[:, :, 1].double().sum(1)
    assert float((s1 - o.sum(1)).abs().max()) < 1e-3 * float(o.abs().sum(1).max())
    assert float(((s2 - (o * o).sum(1)) / (o * o).sum(1)).abs().max()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (32768, 4096, 128)])
def test_gelu_epilogue_is_erf_gelu_at_half_precision(cuda, M, N, K):
    """The epilogue's GELU (2 ^ P5(|x|) form of x Phi(x), csrc/cpx_gemm.hip: gelu_erf) against float64 erf-GELU of the
    SAME float32 pre-activation (the f32 epilogue of the same GEMM): after rounding to bf16 the two agree except where the
    exact value sits within ~1e-6 of a rounding boundary -- never more than 1 bf16 ulp (+ 1e-6 absolute) off, correctly
    rounded in > 99.9 % of the elements, with the negative tail and large arguments included (bias spread over [-9, 9])."""
    g = torch.Generator(device="cpu").manual_seed(3)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = torch.linspace(-9, 9, N).to(cuda)
    z = ops.gemm(A, W, "f32", bias)                                     # exact products, fp32 accumulation, + bias
    got = ops.gemm(A, W, "gelu", bias)
    z64 = z.double()
    exact = 0.5 * z64 * torch.special.erfc(-z64 / 2 ** 0.5)
    err = (got.double() - exact).abs()
    # bf16: 8 significant bits -> ulp(v) = 2^(floor(log2 |v|) - 7).  Never more than one ulp off (+ the approximation's own
    # absolute floor, 6.4e-7, tools/fit_gelu.py: in the far negative tail the exact value is ~1e-18 and only the absolute
    # error means anything) ...
    ulp = 2.0 ** (torch.floor(torch.log2(exact.abs().clamp_min(1e-30))) - 7)
    assert bool((err <= ulp + 1e-6).all()), float((err - ulp).max())
    # ... and almost everywhere it IS the correctly rounded value (half an ulp)
    frac = float((err > 0.5 * ulp * 1.0001 + 1e-6).double().mean())
    assert frac < 1e-3, frac
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 1024), (16384, 1024, 128), (16384, 1024, 256),
                                   (4096, 4096, 4096), (32768, 512, 192 * 2)])
def test_gemm256_epilogues(cuda, M, N, K):
    """the 256x256 8-phase kernel (taken when M%256 == N%256 == 0, K/64 even, >= 256 tiles)
    against the 128x128 kernel and torch fp32"""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(cuda)
    ref = A.float() @ W.float().T
    outs = {}
    run = lambda: [ops.gemm(A, W, "bf16", None), ops.gemm(A, W, "bf16", bias), ops.gemm(A, W, "gelu", bias),
                   ops.gemm(A, W, "relu", bias), ops.gemm(A, W, "resid", bias, res)]
    outs[1] = run()                                  # the product library
    with _lib.use_debug_library() as L:              # 128^2 kernel forced on the same shapes (debug-build switch)
        L.cpx_gemm_set_big(0)
        try:
            outs[0] = run()
        finally:
            L.cpx_gemm_set_big(1)
        dbg_big = run()
    assert all(torch.equal(a, b) for a, b in zip(outs[1], dbg_big))     # debug build at its defaults == product, bit for bit
    exp = [ref, ref + bias, torch.nn.functional.gelu(ref + bias), torch.relu(ref + bias), ref + bias + res.float()]
    for k, (o1, o0, e) in enumerate(zip(outs[1], outs[0], exp)):
        assert _rel(o1.float(), e) < 5e-3, (k, _rel(o1.float(), e))
        # same fp32 accumulation order per output element up to the k-split -> nearly identical
        assert _rel(o1.float(), o0.float()) < 3e-3, k
    assert torch.equal(outs[1][0], outs[0][0]) or _rel(outs[1][0].float(), outs[0][0].float()) < 1e-3


@pytest.mark.parametrize("M,N,K", [(16384, 2048, 256), (8192, 2048, 1024), (16384, 1024, 512)])
def test_gemm256_epilogue_and_schedule_variants_are_bitwise_equal(cuda, M, N, K):
    """The persistent 256^2 kernel's production choices against the forms they replaced (debug-build switches): the direct-store
    epilogue (v_permlane16_swap -> 16-byte buffer stores) against the LDS-staged rows, and the balanced fragment-read schedule
    (residual + statistics epilogue) against the plain 8-phase one -- K = 256 (four K tiles: only the peeled head and tail tiles run),
    512 and 1024, one and two tiles per workgroup; direct = 2 forces the direct-store epilogue on every non-residual epilogue.  Same accumulation order by construction -> bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(cuda)
    stats = ops.row_stats(A) if K == 1024 else None               # (the folded LayerNorm is defined for the 1024-channel rows)
    colsum = W.float().sum(1).contiguous() if K == 1024 else None

    def run():
        o = [ops.gemm(A, W, "bf16", bias), ops.gemm(A, W, "gelu", bias), ops.gemm(A, W, "relu", None)]
        o += list(ops.gemm_ln(A, W, "resid", bias, res, want_stats=True)) if N == 1024 else [ops.gemm(A, W, "resid", bias, res)]
        if stats is not None:
            o.append(ops.gemm_ln(A, W, "gelu", bias, None, ln_stats=stats, ln_colsum=colsum))
        return o

    prod = run()
    with _lib.use_debug_library() as L:
        outs = {}
        try:
            for direct, bal in ((1, 1), (0, 0), (2, 1), (0, 1), (2, 0)):
                L.cpx_gemm_set_direct(direct); L.cpx_gemm_set_balanced(bal)
                outs[direct, bal] = run()
        finally:
            L.cpx_gemm_set_direct(1); L.cpx_gemm_set_balanced(1)
    for key, o in outs.items():
        assert len(o) == len(prod)
        for k, (a, b) in enumerate(zip(prod, o)):
            assert torch.equal(a, b), (key, k)
    ref = torch.nn.functional.gelu(A.float() @ W.float().T + bias)
    assert _rel(prod[1].float(), ref) < 5e-3


@pytest.mark.parametrize("N,K,epi", [(256, 1024, "bf16"), (1024, 256, "gelu")])
def test_gemm_two_gib_operand_or_output_takes_64_bit_addresses(cuda, N, K, epi):
    """The persistent 256^2 kernel addresses its operands -- and the direct-store epilogue its output -- through 32-bit buffer offsets.
    An OPERAND of >= 2^31 bytes (A = [2^20, 1024] bf16 is exactly 2 GiB) must fall back to the one-workgroup-per-tile kernel, an OUTPUT of
    >= 2^31 bytes ([2^20, 1024] from K = 256) to the staged epilogue with its 64-bit store addresses: rows at both ends and across the
    2^31-byte boundary are compared with torch."""
    M = 1 << 20
    g = torch.Generator(device=cuda).manual_seed(3)
    A = torch.randn(M, K, generator=g, device=cuda, dtype=torch.float32).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=cuda) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=cuda)
    assert max(A.numel(), M * N) * 2 == 1 << 31
    out = ops.gemm(A, W, epi, bias)
    rows = torch.cat([torch.arange(0, 512), torch.arange(M // 2 - 256, M // 2 + 256), torch.arange(M - 512, M)]).to(cuda)
    ref = A[rows].float() @ W.float().T + bias
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    assert _rel(out[rows].float(), ref) < 5e-3
    assert bool(torch.isfinite(out[::4097].float()).all())


@pytest.mark.parametrize("M,N", [(8192, 2048), (16384, 4096), (32768, 1024)])
def test_gemm_lin1_one_wave_per_simd_equals_the_eight_wave_kernel(cuda, M, N):
    """PRODUCTION mlp.lin1 (bf16: folded LayerNorm + bias + erf-GELU) runs on csrc/cpx_gemm4w.hip (one wave per SIMD, packed-f32 epilogue);
    cpx_gemm_set_4w(0) in the debug build sends the same call to k_gemm256p.  Same accumulation order and the same epilogue arithmetic
    operation by operation -> bit for bit, one / two / four tiles per workgroup, biases spread so that both GELU tails are exercised."""
    K = 1024
    g = torch.Generator(device="cpu").manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * 3 + 0.5).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = (torch.randn(N, generator=g) * 3).to(cuda)
    stats = ops.row_stats(A)
    colsum = W.float().sum(1).contiguous()
    prod = ops.gemm_ln(A, W, "gelu", bias, None, ln_stats=stats, ln_colsum=colsum)
    with _lib.use_debug_library() as L:
        try:
            L.cpx_gemm_set_4w(1)
            four = ops.gemm_ln(A, W, "gelu", bias, None, ln_stats=stats, ln_colsum=colsum)
            L.cpx_gemm_set_4w(0)
            eight = ops.gemm_ln(A, W, "gelu", bias, None, ln_stats=stats, ln_colsum=colsum)
        finally:
            L.cpx_gemm_set_4w(1)
    assert torch.equal(prod, four) and torch.equal(four, eight)
    xn = torch.nn.functional.layer_norm(A[:512].float(), (K,), eps=1e-6)
    ref = torch.nn.functional.gelu(xn @ W.float().T + bias)
    assert _rel(prod[:512].float(), ref) < 6e-3


@pytest.mark.parametrize("M,K", [(65536, 256), (65536, 1024), (131072, 512), (65536, 4096)])
def test_gemm_resid_stats_one_wave_per_simd_equals_the_eight_wave_kernel(cuda, M, K):
    """DEBUG-BUILD ONLY path (cpx_gemm_set_4w(3); the product keeps attn.proj / mlp.lin2 on k_gemm256p, default switch value 1): bias,
    residual add with the reference's double rounding, partial LayerNorm statistics of the
    output rows on csrc/cpx_gemm4w.hip against k_gemm256p<RESID, STATS | BAL> (cpx_gemm_set_4w(0), debug build): outputs AND statistics bit
    for bit -- the 32-chunk summation tree of the staged epilogue is rebuilt from lane-row swaps, registers and one LDS hand-over --, one
    and two tiles per workgroup, K = 256 (only the peeled K tiles run) to 4096, and in place (out == residual, as the engine calls it)."""
    N = 1024
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    res = (torch.randn(M, N, generator=g) * 2).to(torch.bfloat16).to(cuda)
    prod, pst = ops.gemm_ln(A, W, "resid", bias, res, want_stats=True)
    with _lib.use_debug_library() as L:
        try:
            L.cpx_gemm_set_4w(3)
            four, fst = ops.gemm_ln(A, W, "resid", bias, res, want_stats=True)
            x = res.clone()                                                   # in place
            st = torch.zeros((M, 4, 2), dtype=torch.float32, device=cuda)
            _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["resid"], bias.data_ptr(), x.data_ptr(), x.data_ptr(), N,
                                     None, None, st.data_ptr(), torch.cuda.current_stream().cuda_stream), "gemm_ln in place")
            L.cpx_gemm_set_4w(0)
            eight, est = ops.gemm_ln(A, W, "resid", bias, res, want_stats=True)
        finally:
            L.cpx_gemm_set_4w(1)                                              # the production default (mlp.lin1 only on the one-wave kernel)
    assert torch.equal(prod, eight) and torch.equal(four, eight) and torch.equal(x, eight)
    assert torch.equal(pst, est) and torch.equal(fst, est) and torch.equal(st, est)
    ref = (A[:256].float() @ W.float().T + bias).to(torch.bfloat16).float() + res[:256].float()
    assert _rel(prod[:256].float(), ref) < 5e-3
    o = prod.float()
    assert torch.allclose(pst[:, :, 0].sum(1), o.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(pst[:, :, 1].sum(1), (o * o).sum(1), rtol=1e-4)


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 256), (8192, 2048, 1024), (16384, 2048, 256), (32768, 1024, 1024), (24576, 1024, 512)])
def test_gemm_one_wave_per_simd_prototype_equals_the_production_kernel(cuda, M, N, K):
    """csrc/cpx_gemm4w.hip (debug build): the 256^2 tile with one wave per SIMD (128 x 128 per wave, AGPR accumulators through inline-asm
    MFMAs, one barrier per K tile).  Same accumulation order per output element as k_gemm256p -> bit for bit, K = 256 (only the peeled
    head and tail K tiles run) and 1024."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    ref = ops.gemm(A, W, "bf16", bias)
    out = torch.empty_like(ref)
    with _lib.use_debug_library() as L:
        for _ in range(3):
            _lib.check(L.cpx_gemm4w(A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), N,
                                    torch.cuda.current_stream().cuda_stream), "gemm4w")
            assert torch.equal(out, ref)
    assert _rel(out.float(), A.float() @ W.float().T + bias) < 5e-3


def test_gemm256_identity_asymmetric(cuda):
    """exact layout check of the big kernel: A = [I | 0] rows against an asymmetric W"""
    M, N, K = 16384, 1024, 128
    A = torch.zeros(M, K)
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    A = A.to(torch.bfloat16).to(cuda)
    W = ((torch.arange(N * K).reshape(N, K) * 7) % 251 - 125).to(torch.bfloat16).to(cuda)
    out = ops.gemm(A, W, "bf16", None)
    exp = W.float().T[torch.arange(M, device=cuda) % K]
    assert torch.equal(out.float(), exp)


def test_gemm_identity_asymmetric(cuda):
    """A = I against an asymmetric W catches transposed / permuted fragment maps exactly."""
    K = N = 128
    A = torch.eye(128, K).to(torch.bfloat16).to(cuda)
    W = (torch.arange(N * K).reshape(N, K) % 251 - 125).to(torch.bfloat16).to(cuda)
    out = ops.gemm(A, W, "f32", None)
    assert torch.equal(out, W.float().T.contiguous())


@pytest.mark.parametrize("C", [1024, 256])
def test_layernorm(cuda, C):
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(512, C, generator=g) * 3 + 1).to(torch.bfloat16).to(cuda)
    w = torch.randn(C, generator=g).to(cuda)
    b = torch.randn(C, generator=g).to(cuda)
    out = ops.layernorm(x, w, b, 1e-6)
    ref = torch.nn.functional.layer_norm(x.float(), (C,), w, b, 1e-6)
    assert _rel(out.float(), ref) < 4e-3
    assert float((out.float() - ref).abs().max()) < 0.06


def _attention_ref(qkv, relh63, relw63, nS):
    """flash_forward semantics in fp32 on the half-rounded inputs."""
    B, L, H = nS, 1024, 16
    q, k, v = qkv.float().reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    idx = torch.arange(32)[:, None] - torch.arange(32)[None, :] + 31
    Rh, Rw = relh63[idx.to(relh63.device)], relw63[idx.to(relw63.device)]     # [32,32,64]
    qhw = q.reshape(B, H, 32, 32, 64)
    rel_h = torch.einsum("bnhwc,hkc->bnhwk", qhw, Rh)
    rel_w = torch.einsum("bnhwc,wkc->bnhwk", qhw, Rw)
    bias = (rel_h[..., :, None] + rel_w[..., None, :]).reshape(B, H, L, L)
    att = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias, -1)
    return (att @ v).transpose(1, 2).reshape(B * L, 1024)


def test_attention_relpos(cuda):
    g = torch.Generator().manual_seed(5)
    nS = 2
    qkv = torch.randn(nS * 1024, 3072, generator=g).to(torch.bfloat16).to(cuda)
    relh = (torch.randn(63, 64, generator=g) * 0.3).to(torch.bfloat16)
    relw = (torch.randn(63, 64, generator=g) * 0.3).to(torch.bfloat16)
    pad = lambda t: torch.cat([t.float() * 8, torch.zeros(1, 64)]).to(torch.bfloat16).to(cuda)
    out = ops.attention(qkv, pad(relh), pad(relw))
    ref = _attention_ref(qkv, relh.float().to(cuda), relw.float().to(cuda), nS)
    r = _rel(out.float(), ref)
    assert r < 1e-2, r                      # bf16 P and bf16 output rounding
    assert float((out.float() - ref).abs().max()) < 0.05


def test_attention_spiked_rows(cuda):
    """one key dominates -> the online-softmax rescale path is taken; result ~= that key's V"""
    g = torch.Generator().manual_seed(6)
    qkv = (torch.randn(1024, 3072, generator=g) * 0.1)
    qkv[:, :1024] = 1.0                                       # all q equal
    qkv[700, 1024:2048] = 30.0                                # key 700 spikes for every head
    qkv = qkv.to(torch.bfloat16).to(cuda)
    z = torch.zeros(64, 64, dtype=torch.bfloat16, device=cuda)
    out = ops.attention(qkv, z, z)
    ref = _attention_ref(qkv, torch.zeros(63, 64, device=cuda), torch.zeros(63, 64, device=cuda), 1)
    assert float((out.float() - ref).abs().max()) < 0.02
    assert torch.allclose(out.float()[5], qkv[700, 2048:].float(), atol=0.02)


@pytest.mark.parametrize("depth,nS", [(2, 4), (24, 1)])
def test_net_forward_vs_oracle(cuda, depth, nS):
    sd = synth.make_state_dict(7, None, depth=depth, seed=3)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
    rng = np.random.default_rng(0)
    x = rng.random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5) \
        .reshape(nS * 1024, 192).to(torch.bfloat16).to(cuda)
    L = _lib.lib()
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
    ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=cuda)
    import ctypes as C
    _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream().cuda_stream))
    out = head[:, :640].reshape(nS, 32, 32, 10, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(nS, 10, 256, 256)
    out = out.cpu()
    # oracle channel order: [class logits (7), dY, dX, cellprob]; ours: [flow(3), class(7)]
    ours = torch.cat([out[:, 3:], out[:, :3]], 1)
    ref32 = onet.class_transformer_forward(sd, torch.from_numpy(x))
    sdb = {k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in sd.items()}
    refbf = onet.class_transformer_forward(sdb, torch.from_numpy(x), torch.bfloat16)
    e_ours, e_ref = _rel(ours, ref32), _rel(refbf, ref32)
    print(f"depth {depth}: HIP bf16 vs fp32 oracle {e_ours:.4f}; torch-CPU bf16 vs fp32 {e_ref:.4f}; "
          f"HIP vs torch-CPU bf16 {_rel(ours, refbf):.4f}")
    assert e_ours < 2e-2                       # SURVEY 8c tolerance for half precision
    assert e_ours < 1.5 * e_ref + 2e-3         # no worse than the reference's own bf16 path


def test_engine_end_to_end(cuda):
    """tiles -> ids: network outputs within tolerance of the oracle, ids bit-exact given the
    device's own flow/cellprob/logit tensors."""
    from oracle import classmask, dynamics
    sd = synth.make_state_dict(7, None, depth=2, seed=4)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
    eng = engine.Engine(w, 256, batch_tiles=2)
    tiles = np.stack([synth.render_region(1234, 0, 0, 256, 256), synth.render_region(1234, 224, 448, 256, 256)])
    out = eng.run(torch.from_numpy(tiles).to(cuda))
    fw = onet.make_forward(sd)
    for i in range(2):
        x = tiling.normalize_img(tiles[i:i + 1])
        dP, cp, yc = tiling.run_net(fw, x, batch_size=8)
        assert _rel(out.dP[i].cpu(), torch.from_numpy(dP)) < 2e-2
        assert _rel(out.cellprob[i].cpu(), torch.from_numpy(cp)) < 2e-2
        assert _rel(out.logits[i].cpu(), torch.from_numpy(yc)) < 2e-2
        ref = dynamics.compute_masks(out.dP[i].cpu().numpy(), out.cellprob[i].cpu().numpy())
        assert np.array_equal(ops.masks_to_numpy(out.masks)[i], ref)
        cm, _ = classmask.compute_class_masks(ref, out.logits[i].cpu().numpy())
        assert np.array_equal(out.class_masks[i].cpu().numpy(), cm.astype(np.uint8))
    # flow injection: analytic fields -> known cells, records consistent with the maps
    inj = [synth.analytic_fields(1234, 0, 0, 256, 256, 7), synth.analytic_fields(1234, 224, 448, 256, 256, 7)]
    dPi = torch.from_numpy(np.stack([a[0] for a in inj])).to(cuda)
    cpi = torch.from_numpy(np.stack([a[1] for a in inj])).to(cuda)
    lgi = torch.from_numpy(np.stack([a[2] for a in inj])).to(cuda)
    out = eng.run(torch.from_numpy(tiles).to(cuda), inject=(dPi, cpi, lgi))
    rec = eng.fetch_records(2)
    m = ops.masks_to_numpy(out.masks)
    for i in range(2):
        ref = dynamics.compute_masks(inj[i][0], inj[i][1])
        assert np.array_equal(m[i], ref)
        cm, _ = classmask.compute_class_masks(ref, inj[i][2])
        exp = classmask.instance_records(ref, cm)
        r = rec[rec["tile"] == i]
        assert len(r) == ref.max() >= inj[i][3]
        assert np.array_equal(r["area"], exp["area"]) and np.array_equal(r["cls"], exp["cls"])
        assert np.array_equal(np.stack([r["y0"], r["x0"], r["y1"], r["x1"]], 1), exp["bbox"])
        assert np.array_equal(r["sum_y"], exp["sum_y"]) and np.array_equal(r["sum_x"], exp["sum_x"])


@pytest.mark.parametrize("H,aug,precision", [(256, True, "bf16"), (512, False, "bf16"), (256, False, "fp16"),
                                             (320, True, "bf16")])
def test_engine_variants_tta_512_fp16(cuda, H, aug, precision):
    """configs 4/5 building blocks: --tta (flipped 3x3 / 5x5 sub-tile grids), 512-px tiles (9
    sub-tiles), fp16; network outputs vs the fp32 oracle, ids bit-exact on the device tensors."""
    from oracle import classmask, dynamics
    sd = synth.make_state_dict(7, None, depth=1, seed=9)
    w = engine.NetWeights.from_state_dict(sd, precision, cuda)
    eng = engine.Engine(w, H, batch_tiles=1, augment=aug)
    tile = synth.render_region(4321, 100, 50, H, H)[None]
    out = eng.run(torch.from_numpy(tile).to(cuda))
    fw = onet.make_forward(sd)
    dP, cp, yc = tiling.run_net(fw, tiling.normalize_img(tile), batch_size=8, augment=aug)
    tol = 2e-2 if precision == "bf16" else 5e-3
    assert _rel(out.dP[0].cpu(), torch.from_numpy(dP)) < tol
    assert _rel(out.cellprob[0].cpu(), torch.from_numpy(cp)) < tol
    assert _rel(out.logits[0].cpu(), torch.from_numpy(yc)) < tol
    ref = dynamics.compute_masks(out.dP[0].cpu().numpy(), out.cellprob[0].cpu().numpy())
    assert np.array_equal(ops.masks_to_numpy(out.masks)[0], ref)


@pytest.mark.parametrize("nS", [2, 32])
def test_gemm_qkv_epilogue_writes_v_transposed(cuda, nS):
    """qkv projection epilogue: q|k columns row-major, the V third transposed to [s][head][d][t]
    (nS = 2 takes the 128^2 kernel, nS = 32 the 256^2 kernel)"""
    M = nS * 1024
    g = torch.Generator().manual_seed(nS)
    A = torch.randn(M, 1024, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(3072, 1024, generator=g) / 32).to(torch.bfloat16).to(cuda)
    b = torch.randn(3072, generator=g).to(cuda)
    vT = torch.zeros((nS, 16, 64, 1024), dtype=torch.bfloat16, device=cuda)
    out = torch.zeros((M, 3072), dtype=torch.bfloat16, device=cuda)
    _lib.check(_lib.lib().cpx_gemm_bf16(A.data_ptr(), W.data_ptr(), M, 3072, 1024, ops.EPI["qkv"], b.data_ptr(),
                                        vT.data_ptr(), out.data_ptr(), 3072, torch.cuda.current_stream().cuda_stream))
    ref = ops.gemm(A, W, "bf16", b)
    assert torch.equal(out[:, :2048], ref[:, :2048])
    exp = ref[:, 2048:].reshape(nS, 1024, 16, 64).permute(0, 2, 3, 1)
    assert torch.equal(vT, exp)


@pytest.mark.parametrize("M,N", [(512, 384), (32768, 3072)])
def test_gemm_with_folded_layernorm(cuda, M, N):
    """LN(x) W^T + b  ==  rstd (x W'^T - mean colsum) + b'   (both kernels)"""
    K = 1024
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 2 + 0.3).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / 32).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16).float()
    gamma = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).float()
    beta = (0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).float()
    Wf = (W.float() * gamma[None]).to(torch.bfloat16)
    bf = b + W.float() @ beta
    cs = Wf.float().sum(1)
    st = ops.row_stats(x)
    xf = x.float()
    assert torch.allclose(st[:, 0, 0], xf.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st[:, 0, 1], (xf * xf).sum(1), rtol=1e-5, atol=1e-3)
    out = ops.gemm_ln(x, Wf.to(cuda), "bf16", bf.to(cuda), None, st, cs.to(cuda))
    ref = torch.nn.functional.layer_norm(xf, (K,), gamma.to(cuda), beta.to(cuda), 1e-6) @ W.float().to(cuda).T + b.to(cuda)
    assert _rel(out.float(), ref) < 6e-3, _rel(out.float(), ref)


def test_gemm_resid_emits_row_stats(cuda):
    M, N, K = 32768 * 2, 1024, 1024
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    W = (torch.randn(N, K, generator=g) / 32).to(torch.bfloat16).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(cuda)
    out, st = ops.gemm_ln(A, W, "resid", b, res, want_stats=True)
    o = out.float()
    assert torch.allclose(st[..., 0].sum(1), o.sum(1), rtol=1e-4, atol=2e-2)
    assert torch.allclose(st[..., 1].sum(1), (o * o).sum(1), rtol=1e-4, atol=2e-2)
    assert torch.equal(out, ops.gemm(A, W, "resid", b, res))


@pytest.mark.parametrize("fuse", [True, False])
def test_net_forward_fused_vs_unfused_layernorm(cuda, fuse):
    sd = synth.make_state_dict(7, None, depth=2, seed=5)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda, fuse_ln=fuse)
    nS = 32
    x = np.random.default_rng(1).random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5) \
        .reshape(nS * 1024, 192).to(torch.bfloat16).to(cuda)
    L = _lib.lib()
    import ctypes as C
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
    ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=cuda)
    _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream().cuda_stream))
    out = head[:1024 * 2, :640].reshape(2, 32, 32, 10, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(2, 10, 256, 256).cpu()
    ours = torch.cat([out[:, 3:], out[:, :3]], 1)
    ref32 = onet.class_transformer_forward(sd, torch.from_numpy(x[:2]))
    assert _rel(ours, ref32) < 2e-2, _rel(ours, ref32)


@pytest.mark.parametrize("fts,nS", [([64, 128], 2), ([32], 1), ([64, 128], 8), ([20, 36], 2), ([12, 20, 36, 68], 1),
                                    ([64, 128, 256, 512], 1)])
def test_unet_semantic_head_vs_oracle(cuda, fts, nS):
    """feature_transformation_structure checkpoints: UNet head as conv list on the MFMA GEMM -- any ``n_channels`` list
    the reference's UNet accepts (unet.py:121-196): channel counts that are not multiples of 8 (zero-padded on the host),
    up to the 4 encoder levels the 32 x 32 token grid allows, and the reference's own default [64, 128, 256, 512]"""
    sd = synth.make_state_dict(7, fts, depth=1, seed=11)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
    assert w.c.n_unet_ops == 3 * (2 * len(fts) + 2)
    x = np.random.default_rng(2).random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5) \
        .reshape(nS * 1024, 192).to(torch.bfloat16).to(cuda)
    L = _lib.lib()
    import ctypes as C
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
    nbytes = L.cpx_net_workspace_bytes(nS, w.c.dtype) + L.cpx_unet_workspace_bytes(w.c.unet_ops, w.c.n_unet_ops, nS, w.c.dtype)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream().cuda_stream))
    out = head[:, :640].reshape(nS, 32, 32, 10, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(nS, 10, 256, 256).cpu()
    ours = torch.cat([out[:, 3:], out[:, :3]], 1)
    ref32 = onet.class_transformer_forward(sd, torch.from_numpy(x))
    assert _rel(ours[:, 7:], ref32[:, 7:]) < 2e-2           # flow head unchanged
    assert _rel(ours[:, :7], ref32[:, :7]) < 2e-2, _rel(ours[:, :7], ref32[:, :7])   # UNet class logits


def test_engine_with_unet_head(cuda):
    from oracle import dynamics
    sd = synth.make_state_dict(7, [64, 128], depth=1, seed=12)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
    eng = engine.Engine(w, 256, batch_tiles=2)
    tiles = np.stack([synth.render_region(99, 0, 0, 256, 256), synth.render_region(99, 300, 300, 256, 256)])
    out = eng.run(torch.from_numpy(tiles).to(cuda))
    fw = onet.make_forward(sd)
    for i in range(2):
        dP, cp, yc = tiling.run_net(fw, tiling.normalize_img(tiles[i:i + 1]), batch_size=8)
        assert _rel(out.logits[i].cpu(), torch.from_numpy(yc)) < 2e-2
        assert _rel(out.dP[i].cpu(), torch.from_numpy(dP)) < 2e-2
        ref = dynamics.compute_masks(out.dP[i].cpu().numpy(), out.cellprob[i].cpu().numpy())
        assert np.array_equal(ops.masks_to_numpy(out.masks)[i], ref)


def test_engine_full_batch_is_order_and_slot_invariant(cuda):
    """BASELINE config[1] batch (8 WSI tiles = 32 sub-tiles, all 24 layers): a tile's outputs are bitwise
    independent of its position in the batch, of its batch mates and of the pipeline slot -- the
    size-independent property behind static tile sharding (any rank / batch composition gives the same ids)"""
    sd = synth.make_state_dict(7, None, depth=24, seed=5)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
    eng = engine.Engine(w, 256, batch_tiles=8)
    tiles = np.stack([synth.render_region(77, 224 * i, 224 * (i % 3), 256, 256) for i in range(8)])
    t = torch.from_numpy(tiles).to(cuda)
    a = eng.run(t)
    a_dp, a_cp, a_lg, a_m = a.dP.clone(), a.cellprob.clone(), a.logits.clone(), a.masks.clone()
    perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4], device=cuda)
    b = eng.run(t[perm].contiguous())                           # other slot, permuted positions
    assert torch.equal(b.dP, a_dp[perm]) and torch.equal(b.cellprob, a_cp[perm])
    assert torch.equal(b.logits, a_lg[perm]) and torch.equal(b.masks, a_m[perm])
    c = eng.run(t[:3].contiguous())                             # partial batch: same tiles, fewer mates
    assert torch.equal(c.dP, a_dp[:3]) and torch.equal(c.masks, a_m[:3])
    eng1 = engine.Engine(w, 256, batch_tiles=1)                 # one tile per launch: small-tile GEMM kernels
    d = eng1.run(t[4:5].contiguous())
    assert float((d.dP - a_dp[4:5]).abs().max()) < 0.05 * float(a_dp.abs().max())   # other kernel variant: tolerance only


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_benched_path_depth24_32_subtiles_vs_oracle(cuda, precision):
    """The configuration bench.py times -- depth 24, 32 sub-tiles per launch, fuse_ln=True, i.e. the 256^2
    k_gemm256 kernels with the folded LayerNorm, the RESID statistics epilogue and the QKV / V^T epilogue in
    every layer -- against the fp32 oracle on 3 of the 32 sub-tiles (first, middle, last), at the SURVEY 8c
    half-precision tolerance (rel-L2 <= 2e-2) and within 1.5x the error of the reference's own torch path
    run in the same dtype."""
    L = _lib.lib()
    nS, depth = 32, 24
    assert L.cpx_gemm_uses_big_tile(nS * 1024, 3072, 1024, ops.EPI["qkv"]) and L.cpx_gemm_uses_big_tile(nS * 1024, 1024, 4096, ops.EPI["resid"])
    sd = synth.make_state_dict(7, None, depth=depth, seed=3)
    w = engine.NetWeights.from_state_dict(sd, precision, cuda, fuse_ln=True)
    assert w.c.fuse_ln == 1
    hd = engine.HALF_DTYPES[precision]
    x = np.random.default_rng(0).random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5) \
        .reshape(nS * 1024, 192).to(hd).to(cuda)
    import ctypes as C
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
    ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=cuda)
    _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream().cuda_stream))
    out = head[:, :640].reshape(nS, 32, 32, 10, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(nS, 10, 256, 256)
    pick = [0, 17, 31]
    ours = torch.cat([out[pick][:, 3:], out[pick][:, :3]], 1).cpu()
    xs = torch.from_numpy(x[pick])
    ref32 = onet.class_transformer_forward(sd, xs)
    sdh = {k: v.to(hd) if v.is_floating_point() else v for k, v in sd.items()}
    refh = onet.class_transformer_forward(sdh, xs[:1], hd)             # the reference's own half-precision path (1 sub-tile)
    e_ours, e_ref = _rel(ours, ref32), _rel(refh, ref32[:1])
    per = [_rel(ours[i], ref32[i]) for i in range(len(pick))]
    print(f"{precision} depth 24 / 32 sub-tiles (256^2 fused-LN kernels): rel-L2 vs fp32 oracle {e_ours:.4f} "
          f"(per sub-tile {per}); torch-CPU {precision} vs fp32 {e_ref:.4f}; max-abs {float((ours - ref32).abs().max()):.4f}")
    assert e_ours < 2e-2
    assert e_ours < 1.5 * e_ref + 2e-3


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_net_forward_is_bitwise_independent_of_the_round5_switches(cuda, precision):
    """Round 5 changed HOW a forward runs, not what it computes: mlp.lin1 on the one-wave-per-SIMD kernel (bf16 and fp16 instantiations) and the
    MLP in row parts of 16 384 tokens, the qkv projection's outputs by non-temporal stores; round 6 added the agent-scope (written-through) stores of
    the residual epilogues (cpx_gemm_set_nt bit 3) and of mlp.lin1's (cpx_gemm4w_set_variant(4096) = the ordinary stores of round 5).  The whole head
    tensor of a 32-sub-tile, 3-block forward is bit for bit the same with all switched off (cpx_gemm_set_4w(0), cpx_net_set_mlp_parts(0),
    cpx_gemm_set_nt(0), cpx_gemm4w_set_variant(4096): the round-4 path), in mixed settings, and all on (production)."""
    import ctypes as C
    nS, depth = 32, 3
    sd = synth.make_state_dict(7, None, depth=depth, seed=5)
    hd = engine.HALF_DTYPES[precision]
    x = np.random.default_rng(1).random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5).reshape(nS * 1024, 192).to(hd).to(cuda)

    def forward(L):
        w = engine.NetWeights.from_state_dict(sd, precision, cuda, fuse_ln=True)
        head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
        ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=cuda)
        _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(), ws.numel(),
                                     torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return head
    prod = forward(_lib.lib())
    assert bool(torch.isfinite(prod).all())
    assert _lib.lib().cpx_net_mlp_parts(nS, w_dtype := _lib.DTYPE_CODE[precision]) == 2 and _lib.lib().cpx_net_mlp_parts(16, w_dtype) == 1
    with _lib.use_debug_library() as L:
        try:
            for g4, parts, nt, v4 in ((0, 0, 0, 4096), (1, 0, 1, 4096), (0, 1, 6, 0), (1, 1, 8, 4096), (1, 1, 7, 0), (1, 1, 15, 8192), (1, 1, 15, 0)):
                L.cpx_gemm_set_4w(g4); L.cpx_net_set_mlp_parts(parts); L.cpx_gemm_set_nt(nt); L.cpx_gemm4w_set_variant(v4)
                assert torch.equal(forward(L), prod), (g4, parts, nt, v4)
        finally:
            L.cpx_gemm_set_4w(1); L.cpx_net_set_mlp_parts(1); L.cpx_gemm_set_nt(15); L.cpx_gemm4w_set_variant(0)


def test_engine_fp16_512px_vs_oracle(cuda):
    """BASELINE configs[4] geometry on one GPU: Cellpose-SAM backbone + semantic head in fp16 on 512-px
    tiles (9 sub-tiles each), depth 2: network outputs vs the fp32 oracle, ids bit-exact on the device tensors"""
    from oracle import classmask, dynamics
    sd = synth.make_state_dict(7, None, depth=2, seed=19)
    w = engine.NetWeights.from_state_dict(sd, "fp16", cuda)
    eng = engine.Engine(w, 512, batch_tiles=2)
    assert eng.n_sub == 9
    tiles = np.stack([synth.render_region(4321, 100, 50, 512, 512), synth.render_region(4321, 900, 700, 512, 512)])
    out = eng.run(torch.from_numpy(tiles).to(cuda))
    fw = onet.make_forward(sd)
    for i in range(2):
        dP, cp, yc = tiling.run_net(fw, tiling.normalize_img(tiles[i:i + 1]), batch_size=8)
        for a, b, name in ((out.dP[i], dP, "dP"), (out.cellprob[i], cp, "cellprob"), (out.logits[i], yc, "logits")):
            assert _rel(a.cpu(), torch.from_numpy(b)) < 5e-3, name
        ref = dynamics.compute_masks(out.dP[i].cpu().numpy(), out.cellprob[i].cpu().numpy())
        assert np.array_equal(ops.masks_to_numpy(out.masks)[i], ref)
        cm, _ = classmask.compute_class_masks(ref, out.logits[i].cpu().numpy())
        assert np.array_equal(out.class_masks[i].cpu().numpy(), cm.astype(np.uint8))


# ---- vectors minted by the reference's own functions (tests/golden/make_golden_network.py) -----------------
def _gold_net():
    import os, sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if d not in sys.path:
        sys.path.insert(0, d)
    import make_golden_network as mgn
    return mgn, np.load(os.path.join(d, "reference_network.npz"))


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("fp16", 3e-3), ("bf16", 1.5e-2)])
def test_hip_attention_block_equals_reference_flash_forward(cuda, precision, tol):
    """qkv GEMM -> rel-pos flash attention -> proj GEMM on the device vs the output of the reference's own
    classpose.vit_sam.flash_forward (vit_sam.py:15-65) on a ViT-L sized attention module"""
    mgn, gold = _gold_net()
    dim, heads, hw, rh, rw, B, seed, stride = (int(v) for v in gold["ff_vitl_cfg"])
    x, p = mgn.attention_case(dim, heads, hw, rh, rw, B, seed)
    assert mgn.checksum(x) == pytest.approx(float(gold["ff_vitl_xsum"]), rel=1e-12)
    dt = engine.NET_DTYPES[precision]
    epi = "f32" if precision == "fp32" else "bf16"
    dev = lambda t: t.to(dt).to(cuda).contiguous()
    table = lambda t: dev(torch.cat([engine.interp_rel_pos(t.to(dt)).float() * 8.0, torch.zeros(1, 64)], 0))
    qkv = ops.gemm(dev(x.reshape(B * hw * hw, dim)), dev(p["qkv.weight"]), epi, p["qkv.bias"].to(dt).float().to(cuda))
    ao = ops.attention(qkv, table(p["rel_pos_h"]), table(p["rel_pos_w"]))
    y = ops.gemm(ao, dev(p["proj.weight"]), epi, p["proj.bias"].to(dt).float().to(cuda))
    got = y.float().reshape(B, hw * hw, dim)[:, ::stride].cpu()
    ref = torch.from_numpy(gold["ff_vitl_y"])
    r = _rel(got, ref)
    print(f"flash_forward golden, {precision}: rel-L2 {r:.3e}, max-abs {float((got - ref).abs().max()):.3e}")
    assert r < tol


def test_hip_tiling_equals_reference_run_net(cuda):
    """normalise -> pad -> sub-tile (+TTA flips) on the device, the golden's elementwise stand-in network, then
    un-augment -> taper average -> crop on the device: bit-identical to what the reference's own
    classpose.core.run_net (core.py:75-231) returned for the same tile"""
    mgn, gold = _gold_net()
    for k in range(int(gold["rn_n"])):
        H, W, aug, ncls, bs, seed = (int(v) for v in gold[f"rn_{k}_cfg"])
        tile = mgn.run_net_tile(H, W, seed)
        sub, til = ops.make_subtiles(torch.from_numpy(tile)[None].to(cuda), 256, bool(aug), 0.1)
        o = mgn.fake_net_outputs(sub.cpu(), ncls)
        dP, cp, lg = ops.blend_subtiles(o[:, ncls:].contiguous().to(cuda), o[:, :ncls].contiguous().to(cuda), til, 1)
        yf = torch.cat([dP[0], cp], 0).permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(yf[::5, ::5], gold[f"rn_{k}_yf"]), k
        assert np.array_equal(lg[0].permute(1, 2, 0).cpu().numpy()[::5, ::5], gold[f"rn_{k}_ycf"]), k
        assert np.array_equal(yf[H // 2], gold[f"rn_{k}_yf_row"]), k



@pytest.mark.parametrize("variant", [2, 0, 1])
def test_attention_variants_repeatable_and_agree(cuda, variant):
    """race screen for the hand-synchronised attention kernels (LDS-DMA ring behind counted vmcnt + raw barriers,
    8-wave ping-pong): 40 launches on 32 sub-tiles under concurrent load must be bitwise identical, every variant
    within bf16 rounding of the float64 reference"""
    g = torch.Generator().manual_seed(11)
    nS = 32
    qkv = (torch.randn(nS * 1024, 3072, generator=g) * 0.7).to(torch.bfloat16).to(cuda)
    rel = (torch.randn(64, 64, generator=g) * 0.8).to(torch.bfloat16).to(cuda)
    rel[63] = 0
    noise = torch.empty((8192, 8192), device=cuda)
    side = torch.cuda.Stream(cuda)
    product = ops.attention(qkv, rel, rel) if variant == 2 else None     # the production kernel in the PRODUCT library
    with _lib.use_debug_library() as L:                  # variants 0 / 1 only exist in the -DCPX_DEBUG build
        L.cpx_attention_set_variant(variant)
        try:
            first = ops.attention(qkv, rel, rel)
            for i in range(40):
                if i % 4 == 0:
                    with torch.cuda.stream(side):               # uneven memory load next to the kernel
                        noise.normal_()
                assert torch.equal(ops.attention(qkv, rel, rel), first), (variant, i)
            side.synchronize()
        finally:
            L.cpx_attention_set_variant(2)
    if product is not None:
        assert torch.equal(product, first)
    q, k, v = qkv[:1024].double().reshape(1024, 3, 16, 64).permute(1, 2, 0, 3)
    idx = (torch.arange(32)[:, None] - torch.arange(32)[None, :] + 31).to(cuda)
    R = rel.double()[idx] / 8
    qhw = q.reshape(16, 32, 32, 64)
    bias = (torch.einsum("nhwc,hkc->nhwk", qhw, R)[..., :, None] + torch.einsum("nhwc,wkc->nhwk", qhw, R)[..., None, :]).reshape(16, 1024, 1024)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias, -1) @ v).transpose(0, 1).reshape(1024, 1024)
    assert _rel(first[:1024].double(), ref) < 4e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv3x3_implicit_gemm(cuda, dtype):
    """cpx_conv3x3 (the neck's 3x3 conv without an im2col buffer): bitwise equal to cpx_gemm on the materialised
    [M][9 C] operand (same K order, same accumulation) and within half-precision rounding of torch's conv2d."""
    g = torch.Generator().manual_seed(5)
    S, Cc, N = 3, 256, 256
    x = torch.randn(S * 1024, Cc, generator=g).to(dtype).to(cuda)
    w4 = (torch.randn(N, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5).to(dtype)
    Wt = w4.permute(0, 2, 3, 1).reshape(N, 9 * Cc).contiguous().to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    for epi, b in (("bf16", None), ("relu", bias)):
        got = ops.conv3x3(x, Wt, epi, b)
        img = x.view(S, 32, 32, Cc)
        pad = torch.zeros((S, 34, 34, Cc), dtype=dtype, device=cuda)
        pad[:, 1:33, 1:33] = img
        col = torch.cat([pad[:, ky:ky + 32, kx:kx + 32] for ky in range(3) for kx in range(3)], -1).reshape(S * 1024, 9 * Cc).contiguous()
        want = ops.gemm(col, Wt, epi, b)
        assert torch.equal(got, want)
        ref = torch.nn.functional.conv2d(img.permute(0, 3, 1, 2).float(), w4.float().to(cuda), b, padding=1)
        if epi == "relu":
            ref = ref.relu()
        ref = ref.permute(0, 2, 3, 1).reshape(S * 1024, N)
        err = (got.float() - ref).abs().max().item()
        assert err < (3e-2 if dtype == torch.bfloat16 else 4e-3), err


def test_benched_path_all_subtiles_vs_reference_style_torch_on_gpu(cuda):
    """The benched configuration against the reference's op sequence (F.linear / layer_norm / SDPA with the materialised
    rel-pos bias / gelu: ``oracle.net.class_transformer_forward``) executed by PyTorch-ROCm ON THE SAME GPU: all 32
    sub-tiles against the float32 run (the CPU oracle can afford three), and, for scale, how the hand-written network
    compares with that eager path in the same precision -- error against float32 and time per 32-sub-tile batch."""
    import ctypes as C
    import time
    L = _lib.lib()
    nS, depth = 32, 24
    sd = synth.make_state_dict(7, None, depth=depth, seed=3)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda, fuse_ln=True)
    x = np.random.default_rng(0).random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5) \
        .reshape(nS * 1024, 192).to(torch.bfloat16).to(cuda)
    head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
    ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream

    def ours_forward():
        _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(), ws.numel(), st))
    ours_forward()
    out = head[:, :640].reshape(nS, 32, 32, 10, 8, 8).permute(0, 3, 1, 4, 2, 5).reshape(nS, 10, 256, 256)
    ours = torch.cat([out[:, 3:], out[:, :3]], 1).clone()
    xg = torch.from_numpy(x).to(cuda)
    sd32 = {k: v.to(cuda) for k, v in sd.items()}
    sdh = {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in sd32.items()}
    with torch.no_grad():
        ref32 = torch.cat([onet.class_transformer_forward(sd32, xg[i:i + 8]) for i in range(0, nS, 8)])
        refh = onet.class_transformer_forward(sdh, xg, torch.bfloat16)
    e_ours = [_rel(ours[i].cpu(), ref32[i].cpu()) for i in range(nS)]
    e_torch = _rel(refh.cpu(), ref32.cpu())

    def t(fn, reps):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    with torch.no_grad():
        ms_torch = t(lambda: onet.class_transformer_forward(sdh, xg, torch.bfloat16), 3)
    ms_ours = t(ours_forward, 10)
    print(f"32 sub-tiles, depth 24, bf16 on one MI355X: hand-written network {ms_ours:.2f} ms, rel-L2 vs torch-GPU fp32 "
          f"max {max(e_ours):.4f} / mean {np.mean(e_ours):.4f}; PyTorch-ROCm eager (reference op sequence) {ms_torch:.2f} ms, "
          f"rel-L2 {e_torch:.4f}; speed-up {ms_torch / ms_ours:.2f}x")
    assert max(e_ours) < 2e-2
    assert np.mean(e_ours) < 1.5 * e_torch + 2e-3


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_round_weights_on_the_device_equals_net_to_dtype(cuda, precision):
    """cpx_round_weights == torch's `.to(torch.bfloat16 | torch.float16)` on the host (what `net.to(dtype)` does to every parameter of the
    reference, models.py:37-69 / predict_wsi.py:659-727) bit for bit: random values over 40 binades, exact ties in both directions, the largest
    finite values and beyond, denormals of the target type, +-0, +-inf; odd lengths (the scalar tail); both output forms."""
    import ctypes as C
    hd = {"bf16": torch.bfloat16, "fp16": torch.float16}[precision]
    dtc = _lib.DTYPE_CODE[precision]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(100003, generator=g) * torch.exp2(torch.randint(-24, 17, (100003,), generator=g).float())
    mant = 8 if precision == "bf16" else 11
    ties = (torch.arange(1, 4097).float() * 2 + 1) * 2.0 ** -(mant + 1) * 2.0 ** 3          # odd multiples of half an ulp at 2^3 .. 2^4: exact ties
    special = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), 65504.0, 65519.9, 65520.0, 70000.0, -70000.0, 3.3895e38, 3.4e38,
                            6e-8, 2.9e-8, 3.1e-8, 1e-40, -1e-40, 5.96e-8, 6.1e-5, 9.2e-41])
    x = torch.cat([x, ties, -ties, special])
    L = _lib.lib()
    xd = x.to(cuda)
    for n in (x.numel(), 5, 3, 1):
        out_h = torch.zeros(n, dtype=hd, device=cuda)
        out_f = torch.zeros(n, dtype=torch.float32, device=cuda)
        s = torch.cuda.current_stream(cuda).cuda_stream
        _lib.check(L.cpx_round_weights(xd.data_ptr(), out_h.data_ptr(), n, dtc, 0, s), "round_weights")
        _lib.check(L.cpx_round_weights(xd.data_ptr(), out_f.data_ptr(), n, dtc, 1, s), "round_weights")
        want = x[:n].to(hd)
        assert torch.equal(out_h.cpu().view(torch.int16), want.view(torch.int16))
        assert torch.equal(out_f.cpu().view(torch.int32), want.float().view(torch.int32))


@pytest.mark.parametrize("precision,N,K", [("bf16", 3072, 1024), ("fp16", 4096, 1024), ("bf16", 7, 100), ("fp16", 64, 4096)])
def test_fold_layernorm_on_the_device(cuda, precision, N, K):
    """cpx_fold_layernorm (LayerNorm folded into attn.qkv / mlp.lin1; the reference runs norm1 / norm2 as separate ops, vit_sam.py:30-33,
    175-176): folded weights bit-identical to the host fold of rounds 2 - 5 (`(w.to(hd).float() * gamma.to(hd).float()).to(hd)`); folded
    bias and row sums bit-identical to the float64 restatement rounded once to float32, and within float32 summation noise of the host fold."""
    hd = {"bf16": torch.bfloat16, "fp16": torch.float16}[precision]
    g = torch.Generator().manual_seed(N + K)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(K, generator=g)
    beta = 0.1 * torch.randn(K, generator=g)
    L = _lib.lib()
    wf = torch.empty((N, K), dtype=hd, device=cuda)
    bf = torch.empty(N, dtype=torch.float32, device=cuda)
    cs = torch.empty(N, dtype=torch.float32, device=cuda)
    dev = [t.to(cuda) for t in (w, b, gamma, beta)]
    _lib.check(L.cpx_fold_layernorm(*[t.data_ptr() for t in dev], N, K, _lib.DTYPE_CODE[precision], wf.data_ptr(), bf.data_ptr(),
                                    cs.data_ptr(), torch.cuda.current_stream(cuda).cuda_stream), "fold_layernorm")
    wq, bq, gq, btq = (t.to(hd).float() for t in (w, b, gamma, beta))
    wf_host = (wq * gq[None, :]).to(hd)
    assert torch.equal(wf.cpu().view(torch.int16), wf_host.view(torch.int16))
    dot64 = (wq.double() * btq.double()[None, :]).sum(1)
    bf64 = (bq + dot64.float())
    cs64 = wf_host.double().sum(1).float()
    assert torch.equal(bf.cpu(), bf64) and torch.equal(cs.cpu(), cs64)
    assert torch.allclose(bf.cpu(), bq + wq @ btq, rtol=1e-5, atol=1e-6) and torch.allclose(cs.cpu(), wf_host.float().sum(1), rtol=1e-5, atol=1e-5)


def test_net_weights_converted_on_the_device_equal_the_host_conversion(cuda):
    """NetWeights.from_state_dict (round 6: upload float32, round / fold on the device) against the host-side conversion it replaced, tensor by
    tensor on a depth-2 synthetic checkpoint in both half precisions and float32: every GEMM operand and every rounded vector bit-identical."""
    sd = synth.make_state_dict(7, depth=2, seed=11)
    for precision in ("bf16", "fp16", "fp32"):
        hd = engine.NET_DTYPES[precision]
        w = engine.NetWeights.from_state_dict(sd, precision, cuda)
        kept = {t.data_ptr(): t for t in w.keep}
        get = lambda p: kept[p].cpu()
        b0 = w.blocks[0]
        assert torch.equal(get(b0.proj_w).view(-1), sd["encoder.blocks.0.attn.proj.weight"].to(hd).reshape(-1))
        assert torch.equal(get(b0.proj_b), sd["encoder.blocks.0.attn.proj.bias"].to(hd).float())
        assert torch.equal(get(b0.fc2_w).view(-1), sd["encoder.blocks.0.mlp.lin2.weight"].to(hd).reshape(-1))
        assert torch.equal(get(w.c.pos).view(-1), sd["encoder.pos_embed"].to(hd).float().reshape(-1))
        assert torch.equal(get(w.c.pe_w).view(-1), sd["encoder.patch_embed.proj.weight"].to(hd).reshape(-1))
        if precision != "fp32":
            wq = sd["encoder.blocks.1.mlp.lin1.weight"].to(hd).float()
            gq = sd["encoder.blocks.1.norm2.weight"].to(hd).float()
            assert torch.equal(get(w.blocks[1].fc1_w), (wq * gq[None, :]).to(hd))
        else:
            assert torch.equal(get(w.blocks[1].fc1_w), sd["encoder.blocks.1.mlp.lin1.weight"])


def test_engine_inject_without_class_logits(cuda):
    """A model without a class head (--model_path cpsam: ncls <= 1) takes inject = (dP, cellprob, None) -- the chain never reads the logits pointer --,
    ids equal the oracle's; the same None on a model WITH a class head is a ValueError, not an AttributeError (round-5 advisor)."""
    from oracle import dynamics
    f = [synth.analytic_fields(5, 224 * i, 0, 256, 256, 7) for i in range(2)]
    dP = torch.from_numpy(np.stack([a[0] for a in f])).to(cuda)
    cp = torch.from_numpy(np.stack([a[1] for a in f])).to(cuda)
    lg = torch.from_numpy(np.stack([a[2] for a in f])).to(cuda)
    tiles = torch.from_numpy(np.stack([synth.render_region(5, 224 * i, 0, 256, 256) for i in range(2)])).to(cuda)
    w1 = engine.NetWeights.from_state_dict(synth.make_state_dict(1, None, depth=1, seed=3), "bf16", cuda)
    assert w1.ncls <= 1
    eng1 = engine.Engine(w1, 256, batch_tiles=2)
    out = eng1.run(tiles, inject=(dP, cp, None))
    m = ops.masks_to_numpy(out.masks)
    for i in range(2):
        assert np.array_equal(m[i], dynamics.compute_masks(f[i][0], f[i][1], flow_threshold=0.4).astype(np.uint16)), i
    assert int(out.class_masks.max()) == 0
    w7 = engine.NetWeights.from_state_dict(synth.make_state_dict(7, None, depth=1, seed=3), "bf16", cuda)
    eng7 = engine.Engine(w7, 256, batch_tiles=2)
    with pytest.raises(ValueError, match="logits may be None only without a class head"):
        eng7.submit(tiles, inject=(dP, cp, None))
    with pytest.raises(ValueError):
        eng7.submit(tiles, inject=(None, cp, lg))
    eng7.run(tiles, inject=(dP, cp, lg))


def test_mlp_row_parts_of_uneven_size_are_bitwise_the_unsplit_forward(cuda):
    """Round 6: any batch of >= 32 sub-tiles runs its MLP in floor(n / 16) row parts of 16 384 rows, the last one with the remainder (until round 5 only
    multiples of 16 sub-tiles were split: the reference's default 1024-px tile -- 25 sub-tiles, 200 per launch -- sent 1.68 GB of hidden activations through
    HBM per layer).  56 sub-tiles -> parts of 16 384 / 16 384 / 24 576 rows: head tensor bit for bit the one of the unsplit forward (cpx_net_set_mlp_parts(0))."""
    import ctypes as C
    nS, depth = 56, 2
    sd = synth.make_state_dict(7, None, depth=depth, seed=9)
    x = np.random.default_rng(2).random((nS, 3, 256, 256)).astype(np.float32)
    patches = torch.from_numpy(x).reshape(nS, 3, 32, 8, 32, 8).permute(0, 2, 4, 1, 3, 5).reshape(nS * 1024, 192).to(torch.bfloat16).to(cuda)

    def forward(L):
        w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
        head = torch.empty((nS * 1024, w.c.ld_head), dtype=torch.float32, device=cuda)
        ws = torch.empty(L.cpx_net_workspace_bytes(nS, w.c.dtype), dtype=torch.uint8, device=cuda)
        _lib.check(L.cpx_net_forward(C.byref(w.c), patches.data_ptr(), nS, head.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return head
    L0 = _lib.lib()
    assert [L0.cpx_net_mlp_parts(n, 0) for n in (16, 31, 32, 36, 56, 144, 200)] == [1, 1, 2, 2, 3, 9, 12] and L0.cpx_net_mlp_parts(200, 2) == 1
    prod = forward(L0)
    assert bool(torch.isfinite(prod).all())
    with _lib.use_debug_library() as L:
        try:
            L.cpx_net_set_mlp_parts(0)
            assert torch.equal(forward(L), prod)
        finally:
            L.cpx_net_set_mlp_parts(1)
