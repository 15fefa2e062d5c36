"""The cellpose-derived half of the oracle cannot be pinned against the wheels (absent from the image), so it is pinned
against INDEPENDENT implementations of the same definitions from libraries that are here, and against frozen outputs of
itself (round-3 review, "harden the unpinnable oracle sections"):

* ``max_pool_nd`` (cellpose.dynamics: separable shifted maxima)      == torch.nn.functional.max_pool2d(k, 1, k // 2)
* ``ndimage.binary_fill_holes`` (what stands in for fill_voids.fill)  == a 4-connected flood of the background from the border
* ``fr_renumber`` (fastremap.renumber)                                == a literal first-appearance pass in raster order
* the fp64 heat diffusion of ``masks_to_flows``                       == the literal torch-double loop of cellpose's
  ``_extend_centers_gpu`` (index_put, gather, ``mean(axis=0)``), to 1e-12 relative, with identical keep / drop decisions
* ``compute_masks`` / ``compute_class_masks`` / ``normalize_img``     == tests/golden/oracle_self_golden.npz (drift alarm)
"""
import os
import sys
from collections import deque

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import classmask, dynamics, tiling   # noqa: E402


def test_max_pool_nd_is_max_pool2d_same_padding():
    g = torch.Generator().manual_seed(0)
    for k in (3, 5):
        for shape in ((1, 37, 53), (4, 11, 11), (2, 64, 40)):
            hi = torch.randint(0, 30, shape, generator=g, dtype=torch.int32)
            ref = torch.nn.functional.max_pool2d(hi.float()[None], k, 1, k // 2)[0]
            assert torch.equal(dynamics.max_pool_nd(hi, k).float(), ref)
            hf = torch.rand(shape, generator=g)
            assert torch.equal(dynamics.max_pool_nd(hf, k), torch.nn.functional.max_pool2d(hf[None], k, 1, k // 2)[0])


def _fill_holes_by_border_flood(m: np.ndarray) -> np.ndarray:
    """fill_voids' definition: everything that a 4-connected walk through background cannot reach from the border is filled"""
    H, W = m.shape
    reach = np.zeros((H, W), bool)
    q = deque()
    for y in range(H):
        for x in (0, W - 1):
            if not m[y, x] and not reach[y, x]:
                reach[y, x] = True; q.append((y, x))
    for x in range(W):
        for y in (0, H - 1):
            if not m[y, x] and not reach[y, x]:
                reach[y, x] = True; q.append((y, x))
    while q:
        y, x = q.popleft()
        for dy, dx in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            yy, xx = y + dy, x + dx
            if 0 <= yy < H and 0 <= xx < W and not m[yy, xx] and not reach[yy, xx]:
                reach[yy, xx] = True; q.append((yy, xx))
    return ~reach


def test_binary_fill_holes_is_the_4_connected_border_flood():
    from scipy import ndimage
    rng = np.random.default_rng(1)
    for _ in range(40):
        h, w = rng.integers(3, 40, 2)
        m = rng.random((h, w)) < rng.uniform(0.3, 0.7)
        if rng.random() < 0.5:                       # rings / diagonal leaks: where 4- and 8-connectivity differ
            m = ndimage.binary_dilation(m, iterations=1) & ~ndimage.binary_erosion(m, iterations=1)
        assert np.array_equal(ndimage.binary_fill_holes(m), _fill_holes_by_border_flood(m))


def test_fr_renumber_is_first_appearance_order():
    rng = np.random.default_rng(2)
    for _ in range(20):
        a = rng.choice(np.array([0, 0, 0, 3, 7, 8, 20, 21, 500, 65535], np.uint16), size=(rng.integers(1, 30), rng.integers(1, 30)))
        lut, nxt, exp = {}, 1, np.zeros(a.shape, a.dtype)
        for i, v in enumerate(a.ravel()):            # C order
            if v == 0:
                continue
            if int(v) not in lut:
                lut[int(v)] = nxt; nxt += 1
            exp.ravel()[i] = lut[int(v)]
        assert np.array_equal(dynamics.fr_renumber(a), exp)


def _extend_centers_torch_double(masks: np.ndarray, centers: np.ndarray, n_iter: int) -> np.ndarray:
    """cellpose.dynamics._extend_centers_gpu / masks_to_flows_gpu, literally, on the CPU in float64"""
    mp = torch.from_numpy(np.pad(masks.astype(np.int64), 1))
    y, x = torch.nonzero(mp, as_tuple=True)
    nby = torch.stack((y, y - 1, y + 1, y, y, y - 1, y - 1, y + 1, y + 1))
    nbx = torch.stack((x, x, x, x - 1, x + 1, x - 1, x + 1, x - 1, x + 1))
    isneighbor = mp[nby, nbx] == mp[nby[0], nbx[0]]
    meds = torch.from_numpy(centers.astype(np.int64)) + 1
    T = torch.zeros(mp.shape, dtype=torch.double)
    for _ in range(n_iter):
        T[meds[:, 0], meds[:, 1]] += 1
        Tn = T[nby, nbx]
        Tn *= isneighbor
        T[nby[0], nbx[0]] = Tn.mean(axis=0)
    return T.numpy()


def test_heat_diffusion_equals_literal_torch_double_loop():
    from make_oracle_self_golden import CASES, case_inputs
    for seed, x0, y0, h, w in CASES[:2]:
        dP, cp, _ = case_inputs(seed, x0, y0, h, w)
        _, st = dynamics.compute_masks(dP, cp, return_stages=True)
        masks = st["masks_seeded"]
        mu, dbg = dynamics.masks_to_flows(masks, return_debug=True)
        T = _extend_centers_torch_double(masks, dbg["centers"], dbg["n_iter"])
        scale = np.abs(T).max()
        assert np.abs(dbg["T"] - T).max() <= 1e-12 * scale          # summation order of the 9-term mean: last-ulp differences only
        # and the decision they feed is the same
        y, x = np.nonzero(np.pad(masks, 1))
        dy, dx = T[y + 1, x] - T[y - 1, x], T[y, x + 1] - T[y, x - 1]
        m2 = np.stack((dy, dx)); m2 /= 1e-60 + (m2 ** 2).sum(0) ** 0.5
        mu2 = np.zeros_like(mu); mu2[:, y - 1, x - 1] = m2
        from scipy import ndimage
        idx = np.arange(1, masks.max() + 1)
        e1 = sum(ndimage.mean((mu[i] - dP[i] / 5.0) ** 2, masks, index=idx) for i in range(2))
        e2 = sum(ndimage.mean((mu2[i] - dP[i] / 5.0) ** 2, masks, index=idx) for i in range(2))
        assert np.array_equal(e1 > 0.4, e2 > 0.4) and np.allclose(e1, e2, rtol=1e-9, atol=1e-12)


def test_oracle_outputs_equal_their_frozen_selves():
    """drift alarm for the restatements no reference fixture can pin (tests/golden/make_oracle_self_golden.py)"""
    from make_oracle_self_golden import CASES, case_inputs
    from classpose_amd import synth
    gold = np.load(os.path.join(ROOT, "tests", "golden", "oracle_self_golden.npz"))
    for k, (seed, x0, y0, h, w) in enumerate(CASES):
        dP, cp, lg = case_inputs(seed, x0, y0, h, w)
        masks, st = dynamics.compute_masks(dP, cp, return_stages=True)
        assert masks.max() >= 5
        assert np.array_equal(masks, gold[f"c{k}_masks"])
        assert np.array_equal(st["masks_seeded"], gold[f"c{k}_seeded"])
        assert np.array_equal(st["masks_flowfiltered"], gold[f"c{k}_flowfiltered"])
        assert np.allclose(st["flow_errors"], gold[f"c{k}_flow_errors"], rtol=1e-12, atol=0)
        assert np.array_equal(st["p_final"][:, ::17], gold[f"c{k}_p_final"])
        cm, _ = classmask.compute_class_masks(masks, lg)
        assert np.array_equal(cm.astype(np.uint8), gold[f"c{k}_class"])
        tile = synth.render_region(seed, x0, y0, w, h)
        assert np.array_equal(tiling.normalize_img(tile[None])[0][::9, ::9], gold[f"c{k}_norm"])
