"""GPU parity (bit-exact): HIP post-processing kernels vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from classpose_amd import ops, synth
from oracle import classmask, cref, dynamics

pytestmark = pytest.mark.gpu


def _fields(kind, H, W, seed):
    rng = np.random.default_rng(seed)
    if kind == "discs":
        dP, cp, lg, _ = synth.analytic_fields(1234 + seed, 37 * seed, 91 * seed, W, H, 7)
    elif kind == "noisy_discs":
        dP, cp, lg, _ = synth.analytic_fields(99 + seed, 10 * seed, 5 * seed, W, H, 7)
        dP = dP + rng.standard_normal(dP.shape).astype(np.float32) * 1.5
        cp = cp + rng.standard_normal(cp.shape).astype(np.float32) * 3
        lg = lg + rng.standard_normal(lg.shape).astype(np.float32)
    else:   # smooth random field: big ragged blobs, exercises big-bbox / removal paths
        from scipy.ndimage import gaussian_filter
        dP = np.stack([gaussian_filter(rng.standard_normal((H, W)), 6) for _ in range(2)]) * 60
        cp = gaussian_filter(rng.standard_normal((H, W)), 5) * 20
        lg = rng.standard_normal((7, H, W))
        dP, cp, lg = dP.astype(np.float32), cp.astype(np.float32), lg.astype(np.float32)
    return dP, cp, lg


CASES = [("discs", 256, 256, 1), ("discs", 200, 312, 2), ("noisy_discs", 256, 256, 3),
         ("random", 128, 160, 4), ("random", 256, 256, 5), ("noisy_discs", 97, 131, 6)]


def _unpack(pf, n_active=None):
    pf = pf.cpu().numpy()
    return pf >> 16, (pf & 0xFFFF).astype(np.int16).astype(np.int64)


@pytest.mark.parametrize("kind,H,W,seed", CASES)
def test_follow_flows_bit_exact(cuda, kind, H, W, seed):
    dP, cp, _ = _fields(kind, H, W, seed)
    pf, fl = ops.follow_flows(torch.from_numpy(dP).to(cuda), torch.from_numpy(cp).to(cuda),
                              niter=200, return_float=True)
    inds = np.nonzero(cp > 0)
    ref = cref.follow_flows(dP * (cp > 0) / 5.0, inds, 200)          # == torch grid_sample loop
    fl = fl.cpu().numpy()[0].reshape(2, H, W)
    assert np.array_equal(fl[:, inds[0], inds[1]], ref)
    py, px = _unpack(pf)
    assert np.array_equal(py[0].reshape(H, W)[inds], ref[0].astype(np.int32))
    assert np.array_equal(px[0].reshape(H, W)[inds], ref[1].astype(np.int32))
    assert np.all(pf.cpu().numpy()[0].reshape(H, W)[cp <= 0] == -1)


@pytest.mark.parametrize("niter", [0, 1, 2, 3, 4, 5, 7, 9, 38, 199, 201, 203])
def test_follow_flows_step_counts_bit_exact(cuda, niter):
    """The Euler loop runs in groups of four steps with the orbit-closure test once per group and per wave
    (k_follow<.., GROUPED>): every remainder (niter % 4), counts below one group, and BOTH parities of the steps that
    remain when a wave's orbits close (discs: fixed points and 2-cycles after ~10 steps) must land where the plain
    loop does.  Reference: cellpose steps_interp through models.py:149-159."""
    for kind, H, W, seed in (("discs", 96, 128, 11), ("noisy_discs", 67, 45, 12)):
        dP, cp, _ = _fields(kind, H, W, seed)
        _, fl = ops.follow_flows(torch.from_numpy(dP).to(cuda), torch.from_numpy(cp).to(cuda), niter=niter, return_float=True)
        inds = np.nonzero(cp > 0)
        ref = cref.follow_flows(dP * (cp > 0) / 5.0, inds, niter)
        assert np.array_equal(fl.cpu().numpy()[0].reshape(2, H, W)[:, inds[0], inds[1]], ref), (kind, niter)


@pytest.mark.parametrize("H,W", [(2, 2), (3, 70), (31, 33), (33, 31), (64, 64), (65, 130)])
def test_follow_flows_long_travel_leaves_the_lds_window(cuda, H, W):
    """A constant flow carries every pixel across the tile (far more than the 16-cell halo of the 32 x 32-cell
    segment's LDS window), a vortex keeps them moving for all 200 steps: the steps of a wave that has left its window
    read the field from memory, the result is the plain loop's.  Sizes around the segment grid (one cell short / over),
    a single segment, and the 2 x 2 minimum."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cases = {"constant": np.stack([np.full((H, W), 3.5, np.float32), np.full((H, W), -4.25, np.float32)]),
             "vortex": np.stack([(xx - W / 2) * 0.4, -(yy - H / 2) * 0.4]).astype(np.float32)}
    cp = np.ones((H, W), np.float32)
    cp[::7, ::5] = -1.0
    for name, dP in cases.items():
        _, fl = ops.follow_flows(torch.from_numpy(dP).to(cuda), torch.from_numpy(cp).to(cuda), niter=200, return_float=True)
        inds = np.nonzero(cp > 0)
        ref = cref.follow_flows(dP * (cp > 0) / 5.0, inds, 200)
        assert np.array_equal(fl.cpu().numpy()[0].reshape(2, H, W)[:, inds[0], inds[1]], ref), name


@pytest.mark.parametrize("H,W", [(256, 256), (97, 131)])
def test_follow_flows_variants_are_bitwise_equal(cuda, H, W):
    """Round 5's Euler loop (32 x 32-cell foreground segments, taps from an LDS window with a wave-uniform choice per step,
    orbit test once per four steps and per wave) against every form it replaced or was measured against (debug-build switch
    cpx_follow_set_lds_window: 0 = round 4's loop, 2 / 9 / 11 = one change at a time, 1 / 5 / 7 = the window with the per-step test
    / per-lane choice), with and without the early exit: end points bit-identical on discs, noisy discs and smooth random
    fields; product library == debug library at its defaults."""
    from classpose_amd import _lib
    tiles = [_fields(k, H, W, s) for k, s in (("discs", 31), ("noisy_discs", 32), ("random", 33))]
    dP = torch.from_numpy(np.stack([t[0] for t in tiles])).to(cuda)
    cp = torch.from_numpy(np.stack([t[1] for t in tiles])).to(cuda)

    def run(L, niter):
        pf = torch.empty((3, H * W), dtype=torch.int32, device=cuda)
        fl = torch.empty((3, 2, H * W), dtype=torch.float32, device=cuda)
        ws = torch.empty(L.cpx_postproc_workspace_bytes(3, H, W), dtype=torch.uint8, device=cuda)
        _lib.check(L.cpx_follow_flows(dP.data_ptr(), cp.data_ptr(), 3, H, W, 0.0, niter, pf.data_ptr(), fl.data_ptr(), ws.data_ptr(),
                                      torch.cuda.current_stream(cuda).cuda_stream), "follow_flows")
        torch.cuda.synchronize(cuda)
        return pf.cpu().numpy(), fl.cpu().numpy()

    product = {n: run(_lib.lib(), n) for n in (200, 37)}
    with _lib.use_debug_library() as L:
        try:
            for n in (200, 37):
                assert all(np.array_equal(a, b) for a, b in zip(product[n], run(L, n)))
                for early in (1, 0):
                    L.cpx_follow_set_early_exit(early)
                    for v in (0, 1, 2, 3, 5, 7, 9, 11):
                        L.cpx_follow_set_lds_window(v)
                        got = run(L, n)
                        assert all(np.array_equal(a, b) for a, b in zip(product[n], got)), (n, early, v)
        finally:
            L.cpx_follow_set_early_exit(1)
            L.cpx_follow_set_lds_window(3)


def test_follow_flows_torch_pin_small(cuda):
    """the literal torch path, not only its C restatement"""
    dP, cp, _ = _fields("noisy_discs", 64, 80, 7)
    _, fl = ops.follow_flows(torch.from_numpy(dP).to(cuda), torch.from_numpy(cp).to(cuda), 50,
                             return_float=True)
    inds = np.nonzero(cp > 0)
    ref = dynamics.follow_flows(dP * (cp > 0) / 5.0, inds, 50).numpy()
    assert np.array_equal(fl.cpu().numpy()[0].reshape(2, 64, 80)[:, inds[0], inds[1]], ref)


def test_flow_errors_across_the_diffusion_kernels_lds_threshold(cuda):
    """The diffusion runs a label in the LDS of a 256-thread workgroup when its padded box has at most 2048 cells (first launch), in the
    147 KB of a 1024-thread workgroup up to 8192 cells (second launch, round 5: two planes) or 18 336 cells (one plane, new values held in
    registers across a barrier), and on the global planes beyond: labels on every side of every limit, old (2944, 3584) and new -- squares of
    30 .. 60, 100 and 150 pixels, a 40 x 70 and a 20 x 150 bar, an L-shape in a 58 x 58 box --
    in one tile, flow errors against the oracle's fp64 diffusion (rtol 1e-12) and the ids the filter keeps.
    Reference: cellpose masks_to_flows_gpu / flow_error through models.py:149-159."""
    H, W = 560, 400
    m = np.zeros((H, W), np.int32)
    boxes = [(5, 5, 50, 50), (5, 70, 52, 52), (5, 135, 53, 53), (5, 200, 54, 54), (5, 265, 56, 56), (5, 330, 58, 58),
             (80, 5, 60, 60), (80, 80, 40, 70), (150, 5, 20, 150), (200, 200, 58, 58), (270, 5, 100, 100), (270, 150, 30, 30), (320, 150, 43, 43)]            # (+ a 150 x 150 square below: a box above 18 336 cells)
    boxes.append((400, 100, 150, 150))
    for lab, (y, x, h, w) in enumerate(boxes, 1):
        m[y:y + h, x:x + w] = lab
    m[200 + 20:200 + 58, 200 + 20:200 + 58] = 0                      # label 10: an L in a 58 x 58 box
    cells = [(h + 2) * (w + 2) for _, _, h, w in boxes]
    assert sum(c <= 2048 for c in cells) >= 2 and sum(2048 < c <= 2944 for c in cells) >= 2 and sum(2944 < c <= 3584 for c in cells) >= 3
    assert sum(3584 < c <= 8192 for c in cells) >= 2 and sum(8192 < c <= 18336 for c in cells) >= 1 and sum(c > 18336 for c in cells) >= 1
    rng = np.random.default_rng(3)
    dP = (rng.standard_normal((2, H, W)) * 2).astype(np.float32)
    want_masks, want_err = dynamics.remove_bad_flow_masks(m, dP, 0.4, return_errors=True)
    masks = torch.from_numpy(m.copy()).to(cuda)[None]
    got_masks, errs = ops.remove_bad_flow_masks(masks, torch.from_numpy(dP).to(cuda)[None], 0.4, return_errors=True)
    e = errs.cpu().numpy()[0][: len(want_err)]
    assert np.allclose(e, want_err, rtol=1e-12, atol=1e-14)
    assert not (np.abs(want_err - 0.4) < 1e-9).any()
    assert np.array_equal(got_masks.cpu().numpy()[0], want_masks.astype(np.int32))


@pytest.mark.parametrize("kind,H,W,seed", CASES)
def test_stagewise_masks_bit_exact(cuda, kind, H, W, seed):
    dP, cp, lg = _fields(kind, H, W, seed)
    ref, st = dynamics.compute_masks(dP, cp, return_stages=True)
    dPd, cpd = torch.from_numpy(dP).to(cuda)[None], torch.from_numpy(cp).to(cuda)[None]
    pf = ops.follow_flows(dPd, cpd, 200)
    masks, nlab = ops.get_masks(pf, H, W, 0.4)
    assert np.array_equal(masks.cpu().numpy()[0], st["masks_seeded"].astype(np.int32)), "get_masks"
    assert int(nlab[0]) == int(st["masks_seeded"].max())
    if st["masks_seeded"].max() > 0:
        masks, errs = ops.remove_bad_flow_masks(masks, dPd, 0.4, return_errors=True)
        e = errs.cpu().numpy()[0][: len(st["flow_errors"])]
        amb = np.abs(st["flow_errors"] - 0.4) < 1e-9
        assert np.allclose(e, st["flow_errors"], rtol=1e-12, atol=1e-14), "flow errors"
        if not amb.any():
            assert np.array_equal(masks.cpu().numpy()[0], st["masks_flowfiltered"].astype(np.int32)), "flow filter"
    masks, nlab = ops.fill_holes_and_remove_small_masks(masks, 15)
    assert np.array_equal(masks.cpu().numpy()[0], ref.astype(np.int32)), "fill/size filter"
    assert int(nlab[0]) == int(ref.max())
    cm = ops.compute_class_masks(masks, torch.from_numpy(lg).to(cuda)[None])
    cm_ref, _ = classmask.compute_class_masks(ref, lg)
    assert np.array_equal(cm.cpu().numpy()[0], cm_ref.astype(np.uint8))


def test_compute_masks_batched_equals_oracle(cuda):
    tiles = [_fields(k, 256, 256, s) for k, s in (("discs", 11), ("noisy_discs", 12), ("random", 13),
                                                   ("discs", 14))]
    dP = torch.from_numpy(np.stack([t[0] for t in tiles])).to(cuda)
    cp = torch.from_numpy(np.stack([t[1] for t in tiles])).to(cuda)
    lg = torch.from_numpy(np.stack([t[2] for t in tiles])).to(cuda)
    m, cm, nlab = ops.compute_masks(dP, cp, lg)
    m = ops.masks_to_numpy(m)
    for i, (a, b, c) in enumerate(tiles):
        ref = dynamics.compute_masks(a, b)
        assert np.array_equal(m[i], ref), i
        cref_, _ = classmask.compute_class_masks(ref, c)
        assert np.array_equal(cm.cpu().numpy()[i], cref_.astype(np.uint8)), i
        assert int(nlab[i]) == ref.max()
    # idempotence / determinism: same inputs -> same ids
    m2, _, _ = ops.compute_masks(dP, cp, lg)
    assert np.array_equal(ops.masks_to_numpy(m2), m)


def test_empty_and_degenerate_tiles(cuda):
    H = W = 64
    z = torch.zeros((2, 2, H, W), device=cuda)
    cp = torch.full((2, H, W), -1.0, device=cuda)
    cp[1, 10:30, 10:30] = 1.0                       # active pixels but zero flow: no seed > 10
    m, cm, nlab = ops.compute_masks(z, cp, None)
    assert ops.masks_to_numpy(m).max() == 0 and int(nlab.max()) == 0
    ref = dynamics.compute_masks(np.zeros((2, H, W), np.float32), cp[1].cpu().numpy())
    assert ref.max() == 0


def _fill_cases():
    rng = np.random.default_rng(5)
    cases = []
    m = np.zeros((64, 64), np.int32)             # ring with a hole + a small cell inside the hole
    yy, xx = np.mgrid[:64, :64]
    r = np.hypot(yy - 30, xx - 30)
    m[(r < 20) & (r > 12)] = 1
    m[(np.hypot(yy - 30, xx - 30) < 4)] = 2     # nested label -> sequential path
    m[50:56, 50:58] = 3
    m[52:54, 52:55] = 0                          # plain hole
    cases.append(m)
    m = np.zeros((96, 120), np.int32)            # many blobs with random holes and gaps in ids
    lab = 1
    for _ in range(40):
        y, x = rng.integers(0, 80), rng.integers(0, 100)
        h, w = rng.integers(3, 16), rng.integers(3, 16)
        m[y:y + h, x:x + w] = lab
        if h > 4 and w > 4:
            m[y + 2:y + h - 2, x + 2:x + w - 2] = 0
        lab += rng.integers(1, 3)
    cases.append(m)
    m = np.zeros((150, 150), np.int32)           # bbox > 64 with a spiral-ish hole -> serial path
    m[10:140, 10:140] = 1
    m[20:130, 20:130] = 0
    m[40:110, 40:110] = 2
    m[60:90, 60:90] = 0
    m[20:130, 70:75] = 1
    cases.append(m)
    m = np.zeros((12, 40), np.int32)             # positional-index quirk of the size filter
    m[1:6, 1:6] = 1; m[1:6, 10:15] = 2; m[1:6, 20:25] = 4; m[1:3, 30:33] = 5
    cases.append(m)
    m = np.ones((20, 20), np.int32)              # no background pixel at all
    m[5:15, 5:15] = 2; m[8:10, 8:10] = 3
    cases.append(m)
    # ---- round 5: boxes of 65 .. 256 pixels are filled by a wave with four rows per lane and four 64-bit words per row (fill_label_big)
    m = np.zeros((300, 420), np.int32)           # holes across word (64-column) and row-block (64-row) boundaries, no label inside a hole
    yy, xx = np.mgrid[:300, :420]
    m[np.hypot(yy - 100, xx - 100) < 95] = 1                       # 189 x 189 disc ...
    m[(np.hypot(yy - 100, xx - 100) < 60) & (np.hypot(yy - 100, xx - 100) > 30)] = 0     # ... with an annular hole around an island of itself
    m[38:44, 60:70] = 0; m[62:66, 62:66] = 0; m[126:130, 126:131] = 0; m[100:101, 5:35] = 0   # small holes; a slit from the border that stops short of the annulus (a bay, not a hole)
    m[5:70, 210:410] = 2                                           # 65 x 200 bar with a row of holes through every word
    m[30:40, 215:405:7] = 0
    m[100:290, 215:290] = 3                                        # 190 x 75: a comb -- deep bays open to the outside stay background
    m[110:280, 225:280:10] = 0; m[100:110, 225:280:10] = 0
    m[120:126, 300:306] = 4                                        # ordinary small labels beside them
    m[140:170, 300:340] = 5; m[150:160, 310:330] = 0
    m[200:290, 300:400] = 6; m[263:265, 300:390] = 0; m[205:262, 363:365] = 0; m[210:250, 310:350] = 0    # 90 x 100 spiral-ish: one real hole, two slits
    cases.append(m)
    m = np.zeros((330, 330), np.int32)           # a 300 x 300 frame (box > 256: sequential path) around mid-size labels with holes
    m[10:310, 10:310] = 1; m[20:300, 20:300] = 0
    m[40:140, 40:240] = 2; m[60:120, 60:220] = 0
    m[160:290, 160:290] = 3; m[200:250, 200:250] = 0
    cases.append(m)
    return cases


@pytest.mark.parametrize("idx", range(7))
def test_fill_holes_and_remove_small_masks(cuda, idx):
    m = _fill_cases()[idx]
    ref = dynamics.fill_holes_and_remove_small_masks(m.astype(np.uint16), 15)
    out, nlab = ops.fill_holes_and_remove_small_masks(torch.from_numpy(m.copy()).to(cuda)[None], 15)
    assert np.array_equal(out.cpu().numpy()[0], ref.astype(np.int32))
    assert int(nlab[0]) == ref.max()


def test_compute_class_masks_golden(cuda, golden):
    npz, _ = golden
    for i in range(int(npz["ccm_n"])):
        masks = torch.from_numpy(npz[f"ccm_masks_{i}"].astype(np.int32)).to(cuda)[None]
        lg = torch.from_numpy(npz[f"ccm_logits_{i}"][:, 0]).to(cuda)[None]
        cm = ops.compute_class_masks(masks, lg)
        assert np.array_equal(cm.cpu().numpy()[0].astype(np.int64), npz[f"ccm_out_{i}"]), i


def test_remove_border_instances_golden(cuda, golden):
    npz, _ = golden
    for i in range(int(npz["rbi_n"])):
        a = npz[f"rbi_in_{i}"]
        exp = npz[f"rbi_out_{i}"]
        if a.ndim == 2:
            out = ops.remove_border_instances(torch.from_numpy(a.astype(np.int32)).to(cuda)[None])
            assert np.array_equal(out.cpu().numpy()[0], exp), i
        else:
            inst = torch.from_numpy(a[..., 0].astype(np.int32)).to(cuda)[None]
            cls = torch.from_numpy(a[..., 1].astype(np.uint8)).to(cuda)[None]
            o1, o2 = ops.remove_border_instances(inst, cls)
            assert np.array_equal(o1.cpu().numpy()[0], exp[..., 0]), i
            assert np.array_equal(o2.cpu().numpy()[0], exp[..., 1]), i


def test_compute_masks_maximum_tile_1024(cuda):
    """the CLI's default tile (1024 x 1024): ~1300 cells, label tables of 95k entries"""
    dP, cp, lg = _fields("noisy_discs", 1024, 1024, 21)
    ref = dynamics.compute_masks(dP, cp)
    m, cm, nlab = ops.compute_masks(torch.from_numpy(dP).to(cuda)[None], torch.from_numpy(cp).to(cuda)[None],
                                    torch.from_numpy(lg).to(cuda)[None])
    assert ref.max() > 500
    assert np.array_equal(ops.masks_to_numpy(m)[0], ref)
    cref_, _ = classmask.compute_class_masks(ref, lg)
    assert np.array_equal(cm.cpu().numpy()[0], cref_.astype(np.uint8))


@pytest.mark.parametrize("H,W", [(61, 53), (97, 131), (40, 203)])
def test_compute_masks_odd_sizes_batched(cuda, H, W):
    """odd H, W and H*W not a multiple of 8: 8-pixel runs straddle image rows, tile bases are not 16-byte
    aligned, the bordered flow field has an odd row pitch -- ids, classes and records still equal the oracle"""
    tiles = [_fields(k, H, W, s) for k, s in (("discs", 3), ("noisy_discs", 4), ("random", 5))]
    dP = torch.from_numpy(np.stack([t[0] for t in tiles])).to(cuda)
    cp = torch.from_numpy(np.stack([t[1] for t in tiles])).to(cuda)
    lg = torch.from_numpy(np.stack([t[2] for t in tiles])).to(cuda)
    m, cm, nlab = ops.compute_masks(dP, cp, lg)
    mh = ops.masks_to_numpy(m)
    for i, (a, b, c) in enumerate(tiles):
        ref = dynamics.compute_masks(a, b)
        assert np.array_equal(mh[i], ref), i
        cref_, _ = classmask.compute_class_masks(ref, c)
        assert np.array_equal(cm.cpu().numpy()[i], cref_.astype(np.uint8)), i
        assert int(nlab[i]) == ref.max()


def test_compute_masks_repeatable_under_load(cuda):
    """Race screen for the post-processing chain (LDS-aggregated atomics, compacted diffusion, wave-per-seed growth,
    deferred removals): 25 runs on the same 8-tile batch while a GEMM stream keeps the CUs busy must give
    bit-identical id maps, class maps and label counts, and equal the oracle."""
    from classpose_amd import synth
    f = [synth.analytic_fields(4321, 224 * i, 448, 256, 256, 7) for i in range(8)]
    dP, cp, lg = (torch.from_numpy(np.stack([a[k] for a in f])).to(cuda) for k in range(3))
    side = torch.cuda.Stream(device=cuda)
    g = torch.Generator().manual_seed(0)
    A = torch.randn(8192, 1024, generator=g).to(torch.bfloat16).to(cuda)
    W = torch.randn(4096, 1024, generator=g).to(torch.bfloat16).to(cuda)
    first = None
    for it in range(25):
        with torch.cuda.stream(side):
            for _ in range(2):
                ops.gemm(A, W, "gelu")
        m, cm, nl = ops.compute_masks(dP, cp, lg)
        cur = (m.clone(), cm.clone(), nl.clone())
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, first)), f"run {it} differs"
    torch.cuda.synchronize()
    want = dynamics.compute_masks(f[3][0], f[3][1])
    assert np.array_equal(ops.masks_to_numpy(first[0][3]), want.astype(np.uint16))


def test_compute_masks_with_large_instances_repeatable_under_load_and_equal_to_the_oracle(cuda):
    """Race screen for round 5's large-instance paths -- the diffusion's second launch (two-plane and one-plane LDS forms, global
    planes), the four-rows-per-lane hole fill, the LDS window of the Euler loop with positions that leave it: tiles of discs of radius
    20 / 34 / 50 / 70 (boxes of 41 .. 141 pixels) with pinholes punched into the cell probability, 12 runs under a concurrent GEMM
    stream must give bit-identical id maps and label counts; one tile is checked against the oracle."""
    H = W = 256
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    tiles = []
    for R in (20, 34, 50, 70):
        dP = np.zeros((2, H, W), np.float32); cp = np.full((H, W), -5.0, np.float32)
        step = 2 * R + 5
        for cy in range(R + 2, H - R - 2, step):
            for cx in range(R + 2, W - R - 2, step):
                dy, dx = cy - yy, cx - xx
                r = np.sqrt(dy * dy + dx * dx)
                inside = r <= R
                k = 5.0 / np.maximum(r, 1.0)
                dP[0][inside] = (dy * k)[inside]; dP[1][inside] = (dx * k)[inside]
                cp[inside] = 5.0
                cp[cy - R // 2: cy - R // 2 + 2, cx + 3: cx + 6] = -5.0            # a pinhole: background pixels inside the disc
        tiles.append((dP, cp))
    dP = torch.from_numpy(np.stack([t[0] for t in tiles] * 2)).to(cuda)
    cp = torch.from_numpy(np.stack([t[1] for t in tiles] * 2)).to(cuda)
    side = torch.cuda.Stream(device=cuda)
    g = torch.Generator().manual_seed(0)
    A = torch.randn(8192, 1024, generator=g).to(torch.bfloat16).to(cuda)
    Wt = torch.randn(4096, 1024, generator=g).to(torch.bfloat16).to(cuda)
    first = None
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(2):
                ops.gemm(A, Wt, "gelu")
        m, cm, nl = ops.compute_masks(dP, cp, None)
        cur = (m.clone(), nl.clone())
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, first)), f"run {it} differs"
    torch.cuda.synchronize()
    assert [int(v) for v in first[1][:4]] == [int(v) for v in first[1][4:]]
    for t in (1, 3):
        want = dynamics.compute_masks(tiles[t][0], tiles[t][1])
        assert want.max() >= 1
        assert np.array_equal(ops.masks_to_numpy(first[0][t]), want.astype(np.uint16)), t


def test_compute_masks_equals_reference_eval_golden(cuda):
    """The HIP dynamics + class vote on the network fields of the reference's own ``ClassposeModel.eval`` run
    (tests/golden/reference_eval.npz, minted by make_golden_eval.py; the fields are regenerated with the oracle's
    normalize_img / run_net, which tests/test_oracle_network_pins.py shows bit-identical to the reference's): the id map
    and the class map equal what the reference returned, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_eval as mge
    from oracle import tiling
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_eval.npz"))
    for k in range(int(g["ev_n"])):
        seed, tta, bs = (int(v) for v in g[f"ev_{k}_cfg"])
        x = tiling.normalize_img(mge.eval_tile(seed)[None])

        def fw(img):
            o = mge.decode_net(torch.from_numpy(np.ascontiguousarray(img)), mge.NCLS).numpy()
            return o[:, mge.NCLS:], o[:, :mge.NCLS]
        dP, cp, yc = tiling.run_net(fw, x, batch_size=bs, augment=bool(tta), tile_overlap=0.1, bsize=256)
        assert np.array_equal(dP[:, ::4, ::4], g[f"ev_{k}_dP"])
        m, cm, nl = ops.compute_masks(torch.from_numpy(dP[None]).to(cuda), torch.from_numpy(cp[None]).to(cuda),
                                      torch.from_numpy(yc[None]).to(cuda))
        assert np.array_equal(ops.masks_to_numpy(m)[0], g[f"ev_{k}_masks"])
        assert np.array_equal(cm[0].cpu().numpy(), g[f"ev_{k}_class_masks"])
        assert int(nl[0]) == int(g[f"ev_{k}_masks"].max())


def _chain(L, dP, cp, lg, want_records=True, flow_threshold=0.4):
    """cpx_compute_masks_records through the raw ABI of library L -> (masks u16, class map, nlabels, records, counts, launches)"""
    import ctypes as C
    from classpose_amd import _lib
    nT, _, H, W = dP.shape
    dev = dP.device
    ncls = 0 if lg is None else lg.shape[1]
    max_rec = min(L.cpx_postproc_max_labels(H, W), 65535)
    masks = torch.zeros((nT, H, W), dtype=torch.int16, device=dev)
    cm = torch.full((nT, H, W), 255, dtype=torch.uint8, device=dev)
    nlab = torch.full((nT,), -7, dtype=torch.int32, device=dev)
    rec = torch.zeros(nT * max_rec * C.sizeof(_lib.CpxRecord), dtype=torch.uint8, device=dev)
    cnt = torch.full((nT,), -7, dtype=torch.int32, device=dev)
    ws = torch.empty(L.cpx_postproc_workspace_bytes(nT, H, W), dtype=torch.uint8, device=dev)
    n0 = L.cpx_postproc_launch_count()
    _lib.check(L.cpx_compute_masks_records(dP.data_ptr(), cp.data_ptr(), lg.data_ptr() if lg is not None else None, nT, ncls, H, W,
                                           0.0, flow_threshold, 200, 15, 0.4, masks.data_ptr(), cm.data_ptr(), nlab.data_ptr(),
                                           max_rec if want_records else 0, rec.data_ptr() if want_records else None,
                                           cnt.data_ptr() if want_records else None, ws.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "compute_masks_records")
    launches = int(L.cpx_postproc_launch_count() - n0)
    torch.cuda.synchronize(dev)
    recs = []
    if want_records:
        r = rec.cpu().numpy().view(np.dtype([("tile", "<i4"), ("label", "<i4"), ("cls", "<i4"), ("area", "<i4"), ("y0", "<i4"), ("x0", "<i4"),
                                              ("y1", "<i4"), ("x1", "<i4"), ("sum_y", "<i8"), ("sum_x", "<i8")])).reshape(nT, max_rec)
        recs = [r[t, :int(cnt[t])].copy() for t in range(nT)]
    return ops.masks_to_numpy(masks), cm.cpu().numpy(), nlab.cpu().numpy(), recs, cnt.cpu().numpy(), launches


@pytest.mark.parametrize("vote,flow_thr", [(True, 0.4), (False, 0.4), (True, 0.0)])
def test_fused_chain_equals_stagewise_chain(cuda, vote, flow_thr):
    """The 23-launch fused chain of cpx_compute_masks_records (one initialisation, relabel on the next stage's pass, removals
    through the rank table, histogram in the Euler loop, records in the final pass) against the stage-wise sequence it replaces
    (39 launches; cpx_postproc_set_fused(0) in the debug library): ids, class maps, label counts and per-cell records
    bit-identical on analytic, noisy and random fields incl. an empty tile, both equal to the oracle."""
    from classpose_amd import _lib
    tiles = [_fields(k, 256, 256, s) for k, s in (("discs", 21), ("noisy_discs", 22), ("random", 23), ("discs", 24),
                                                   ("noisy_discs", 25), ("random", 26))]
    dP = torch.from_numpy(np.stack([t[0] for t in tiles] + [np.zeros((2, 256, 256), np.float32)])).to(cuda)
    cp = torch.from_numpy(np.stack([t[1] for t in tiles] + [np.full((256, 256), -1.0, np.float32)])).to(cuda)
    lg = torch.from_numpy(np.stack([t[2] for t in tiles] + [np.zeros((7, 256, 256), np.float32)])).to(cuda) if vote else None
    product = _chain(_lib.lib(), dP, cp, lg, flow_threshold=flow_thr)
    assert product[5] == (23 if vote else 21) - (0 if flow_thr > 0 else 5)
    with _lib.use_debug_library() as L:
        L.cpx_postproc_set_fused(0)
        try:
            staged = _chain(L, dP, cp, lg, flow_threshold=flow_thr)
        finally:
            L.cpx_postproc_set_fused(1)
        fused_dbg = _chain(L, dP, cp, lg, flow_threshold=flow_thr)
    assert staged[5] >= (34 if flow_thr > 0 else 28)
    for got in (product, fused_dbg):
        assert np.array_equal(got[0], staged[0]) and np.array_equal(got[1], staged[1]) and np.array_equal(got[2], staged[2])
        assert np.array_equal(got[4], staged[4])
        for a, b in zip(got[3], staged[3]):
            assert a.tobytes() == b.tobytes()
    for i, (a, b, c) in enumerate(tiles):
        ref = dynamics.compute_masks(a, b, flow_threshold=flow_thr if flow_thr > 0 else None)
        assert np.array_equal(product[0][i], ref), i
        if vote:
            cref_, _ = classmask.compute_class_masks(ref, c)
            assert np.array_equal(product[1][i], cref_.astype(np.uint8)), i
            exp = classmask.instance_records(ref, cref_)
            r = product[3][i]
            assert len(r) == ref.max() and np.array_equal(r["area"], exp["area"]) and np.array_equal(r["cls"], exp["cls"])
            assert np.array_equal(r["sum_y"], exp["sum_y"]) and np.array_equal(r["sum_x"], exp["sum_x"])
        else:
            assert product[1][i].max() == 0
        assert int(product[2][i]) == ref.max()
    assert product[0][6].max() == 0 and int(product[4][6]) == 0 and int(product[2][6]) == 0
    # without records: same maps, the record buffers untouched
    norec = _chain(_lib.lib(), dP, cp, lg, want_records=False, flow_threshold=flow_thr)
    assert np.array_equal(norec[0], product[0]) and np.array_equal(norec[1], product[1]) and int(norec[4][0]) == -7


@pytest.mark.parametrize("flow_thr,grid", [(0.0, 6), (0.4, 6), (0.0, 12)])
def test_fused_chain_with_more_labels_than_the_tail_lds_table(cuda, flow_thr, grid):
    """A dense 512 x 512 tile: sinks on a 6-px grid -> 85 x 85 = 7 225 labels (on a 12-px grid: 1 849), more than the 1 024 the fused
    chain's per-tile tail ranks by comparison through LDS, so its positional ranking runs (round 5: a bitmap of the labels' first pixels and
    the prefix counts of its words instead of every label against every other).  Ids, label counts and records bit-identical to the
    stage-wise chain; a second, sparse tile in the same batch takes the LDS path."""
    from classpose_amd import _lib
    H = W = 512
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cy, cx = (np.floor(yy / grid) * grid + (grid - 1) / 2), (np.floor(xx / grid) * grid + (grid - 1) / 2)
    dense = np.stack([(cy - yy) * 2.5, (cx - xx) * 2.5]).astype(np.float32)
    sparse = _fields("discs", H, W, 41)
    dP = torch.from_numpy(np.stack([dense, sparse[0]])).to(cuda)
    cp = torch.from_numpy(np.stack([np.ones((H, W), np.float32), sparse[1]])).to(cuda)
    product = _chain(_lib.lib(), dP, cp, None, flow_threshold=flow_thr)
    with _lib.use_debug_library() as L:
        L.cpx_postproc_set_fused(0)
        try:
            staged = _chain(L, dP, cp, None, flow_threshold=flow_thr)
        finally:
            L.cpx_postproc_set_fused(1)
    assert int(staged[2][0]) > (6144 if grid == 6 else 1024) and 0 < int(staged[2][1]) < 1024, staged[2]
    assert np.array_equal(product[0], staged[0]) and np.array_equal(product[2], staged[2]) and np.array_equal(product[4], staged[4])
    for a, b in zip(product[3], staged[3]):
        assert a.tobytes() == b.tobytes()
    m = product[0][0].astype(np.int64)
    assert len(np.unique(m[m > 0])) == int(product[2][0])          # ids are 1..n without gaps
    # ... and against the ORACLE (round-5 review: the positional ranking and the LDS-chunk tail had only been compared with another path
    # through the same library): cellpose's get_masks_torch -> flow-error filter -> fill_holes_and_remove_small_masks -> fastremap.renumber
    # restated in oracle/dynamics.py (reference: models.py:149-159), both tiles of the batch, ids and label counts bit for bit
    for t, (a, b) in enumerate([(dense, np.ones((H, W), np.float32)), (sparse[0], sparse[1])]):
        want = dynamics.compute_masks(a, b, flow_threshold=flow_thr if flow_thr > 0 else None)
        assert int(want.max()) == int(product[2][t]), (t, int(want.max()), int(product[2][t]))
        assert np.array_equal(product[0][t], want.astype(np.uint16)), t
        r = product[3][t]
        area = np.bincount(want.ravel().astype(np.int64), minlength=int(want.max()) + 1)[1:]
        assert len(r) == want.max() and np.array_equal(r["label"], np.arange(1, want.max() + 1)) and np.array_equal(r["area"], area)


@pytest.mark.parametrize("R,H", [(67, 256), (100, 512), (180, 512)])
def test_fused_chain_large_box_diffusion_on_global_planes_equals_oracle(cuda, R, H):
    """ONE label whose box (2R + 1 = 135 / 201 / 361 px) exceeds the 133 x 133 px the diffusion's LDS forms hold: inside the FUSED product chain
    (cpx_compute_masks_records, flow_threshold 0.4) its fp64 heat diffusion runs on the global planes (k_diffuse, second launch).  Ids, label
    count and the record against the oracle (cellpose remove_bad_flow_masks / masks_to_flows_gpu restated in oracle/dynamics.py, reference
    call models.py:149-159).  The big disc is off-centre and has a pinhole; small discs sit in the corners, one of them with its flows turned by
    90 degrees so that the filter has a label to remove beside the ones it keeps."""
    from classpose_amd import _lib
    W = H
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    dP = np.zeros((2, H, W), np.float32); cp = np.full((H, W), -5.0, np.float32)

    def disc(cy, cx, rad, rotate=False):
        dy, dx = cy - yy, cx - xx
        r = np.sqrt(dy * dy + dx * dx)
        inside = r <= rad
        k = 5.0 / np.maximum(r, 1.0)
        fy, fx = ((dx * k), (-dy * k)) if rotate else ((dy * k), (dx * k))
        dP[0][inside] = fy[inside]; dP[1][inside] = fx[inside]
        cp[inside] = 5.0
    disc(H // 2 - 3, W // 2 + 2, R)
    cp[H // 2 - R // 2: H // 2 - R // 2 + 2, W // 2 + 5: W // 2 + 9] = -5.0             # a pinhole inside the big disc
    for cy, cx in ((12, 12), (12, W - 14), (H - 13, 13)):
        if (cy - (H // 2 - 3)) ** 2 + (cx - (W // 2 + 2)) ** 2 > (R + 12) ** 2:
            disc(cy, cx, 9)
    if R < 150:
        disc(H - 13, W - 14, 9, rotate=True)                                            # flows turned by 90 degrees: fails the flow-error test
    sparse = _fields("discs", H, W, 17)
    dPb = torch.from_numpy(np.stack([dP, sparse[0]])).to(cuda)
    cpb = torch.from_numpy(np.stack([cp, sparse[1]])).to(cuda)
    got = _chain(_lib.lib(), dPb, cpb, None, flow_threshold=0.4)
    for t, (a, b) in enumerate([(dP, cp), (sparse[0], sparse[1])]):
        want = dynamics.compute_masks(a, b, flow_threshold=0.4)
        assert want.max() >= 1
        assert int(got[2][t]) == int(want.max()) and np.array_equal(got[0][t], want.astype(np.uint16)), t
    big = got[3][0][np.argmax(got[3][0]["area"])]
    assert big["y1"] - big["y0"] + 1 > 133 and big["x1"] - big["x0"] + 1 > 133, big


def test_fused_chain_repeatable_under_load_and_odd_sizes(cuda):
    """race screen of the fused chain (per-stage table copies, relabel-on-load with write-back): 20 runs on an 8-tile batch
    beside a GEMM stream, and odd tile sizes, always the same bits as the stage-wise chain"""
    from classpose_amd import _lib, synth
    f = [synth.analytic_fields(777, 224 * i, 448, 256, 256, 7) for i in range(8)]
    dP, cp, lg = (torch.from_numpy(np.stack([a[k] for a in f])).to(cuda) for k in range(3))
    side = torch.cuda.Stream(device=cuda)
    g = torch.Generator().manual_seed(0)
    A = torch.randn(8192, 1024, generator=g).to(torch.bfloat16).to(cuda)
    Wt = torch.randn(4096, 1024, generator=g).to(torch.bfloat16).to(cuda)
    with _lib.use_debug_library() as L:
        L.cpx_postproc_set_fused(0)
        staged = _chain(L, dP, cp, lg)
        L.cpx_postproc_set_fused(1)
    for it in range(20):
        with torch.cuda.stream(side):
            for _ in range(2):
                ops.gemm(A, Wt, "gelu")
        got = _chain(_lib.lib(), dP, cp, lg)
        assert np.array_equal(got[0], staged[0]) and np.array_equal(got[1], staged[1]) and np.array_equal(got[4], staged[4]), it
        assert all(a.tobytes() == b.tobytes() for a, b in zip(got[3], staged[3])), it
    torch.cuda.synchronize()
    for H, W in ((97, 131), (200, 312), (64, 64)):
        t = [_fields("noisy_discs", H, W, 31), _fields("random", H, W, 32)]
        d, c, l = (torch.from_numpy(np.stack([x[k] for x in t])).to(cuda) for k in range(3))
        with _lib.use_debug_library() as L:
            L.cpx_postproc_set_fused(0)
            st = _chain(L, d, c, l)
            L.cpx_postproc_set_fused(1)
        got = _chain(_lib.lib(), d, c, l)
        assert np.array_equal(got[0], st[0]) and np.array_equal(got[1], st[1]) and np.array_equal(got[2], st[2])
        assert all(a.tobytes() == b.tobytes() for a, b in zip(got[3], st[3]))
        assert np.array_equal(got[0][0], dynamics.compute_masks(t[0][0], t[0][1]))
