"""CPU: oracle/polygons.py (numpy / pure-Python restatement of cv2.findContours(RETR_EXTERNAL,
CHAIN_APPROX_SIMPLE)[0] + shapely ring metrics, predict_wsi.py:601-652) against OpenCV's documented
known answers, and the product's host polygoniser (cpx_polygonize_host, a pure host function of
the C ABI) against that oracle.  The device polygoniser is compared with the same oracle in
tests/test_gpu_polygons.py."""
import numpy as np
import pytest

from classpose_amd import postprocess
from classpose_amd.engine import RECORD_DTYPE
from oracle import polygons as opoly


def _records(m, cm=None):
    """what cpx_instance_records emits; the class is the class map at the instance's first raster pixel when cm is given"""
    labs = [l for l in np.unique(m) if l]
    recs = np.zeros(len(labs), RECORD_DTYPE)
    for i, l in enumerate(labs):
        ys, xs = np.nonzero(m == l)
        cls = 1 + int(l) % 6 if cm is None else int(cm[ys[0], xs[0]])
        recs[i] = (0, l, cls, len(ys), ys.min(), xs.min(), ys.max() + 1, xs.max() + 1, ys.sum(), xs.sum())
    return recs


def _c(mask):
    return [c.tolist() for c in opoly.find_contours_external_simple(np.asarray(mask, np.uint8))]


# ---- known answers of cv2.findContours(..., RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) -------------------------
def test_contour_known_answers():
    m = np.zeros((5, 5), np.uint8); m[1:4, 1:4] = 1
    # the OpenCV tutorial square: starts at the top-left pixel and runs DOWN first (counter-clockwise on screen)
    assert _c(m) == [[[1, 1], [1, 3], [3, 3], [3, 1]]]
    m = np.zeros((6, 9), np.uint8); m[2:5, 1:8] = 1
    assert _c(m) == [[[1, 2], [1, 4], [7, 4], [7, 2]]]
    m = np.zeros((4, 4), np.uint8); m[2, 1] = 1
    assert _c(m) == [[[1, 2]]]                                   # isolated pixel: one point
    m = np.zeros((3, 8), np.uint8); m[1, 2:7] = 1
    assert _c(m) == [[[2, 1], [6, 1]]]                           # 1-px horizontal line: its two ends
    m = np.zeros((8, 3), np.uint8); m[2:7, 1] = 1
    assert _c(m) == [[[1, 2], [1, 6]]]
    m = np.zeros((5, 5), np.uint8); m[2, 1:4] = 1; m[1:4, 2] = 1
    assert _c(m) == [[[2, 1], [1, 2], [2, 3], [3, 2]]]           # 8-connected "plus" -> diamond
    m = np.zeros((5, 5), np.uint8); m[1, 1] = m[2, 2] = m[3, 3] = 1
    assert _c(m) == [[[1, 1], [3, 3]]]                           # diagonal line: 8-connected, two ends
    # L shape: every corner of the pixel-centre outline, downwards first
    m = np.zeros((7, 7), np.uint8); m[1:6, 1:3] = 1; m[4:6, 1:6] = 1
    assert _c(m) == [[[1, 1], [1, 5], [5, 5], [5, 4], [3, 4], [2, 3], [2, 1]]]


def test_contour_order_holes_and_islands():
    m = np.zeros((12, 12), np.uint8)
    m[1:4, 1:4] = 1              # found first
    m[6:11, 2:9] = 1             # found second, has a hole with an island
    m[7:10, 4:7] = 0
    m[8, 5] = 1                  # island inside the hole: not an external contour
    cs = _c(m)
    assert len(cs) == 2
    assert cs[0] == [[2, 6], [2, 10], [8, 10], [8, 6]]           # contours[0] = LAST discovered component
    assert cs[1] == [[1, 1], [1, 3], [3, 3], [3, 1]]
    # the hole does not show up in the external border
    assert all(p not in ([4, 7], [6, 9]) for p in cs[0])


def test_ring_metrics_known_answers():
    sq = np.array([[0, 0], [0, 2], [2, 2], [2, 0]], float)
    r = opoly.polygon_metrics(sq)
    assert r == dict(area=4.0, perimeter=8.0, centroid=(1.0, 1.0), valid=True)
    tri = np.array([[0, 0], [4, 0], [0, 3]], float)              # 3 vertices: the reference drops < 4 before shapely
    assert opoly.polygon_metrics(tri)["area"] == 6.0 and not opoly.polygon_metrics(tri)["valid"]
    lshape = np.array([[0, 0], [0, 4], [4, 4], [4, 3], [1, 3], [1, 0]], float)
    r = opoly.polygon_metrics(lshape)
    assert r["area"] == 7.0 and r["perimeter"] == 16.0 and r["valid"]
    assert np.allclose(r["centroid"], ((4 * 0.5 + 3 * 2.5) / 7 * 1.0, (4 * 2.0 + 3 * 3.5) / 7))
    bow = np.array([[0, 0], [2, 2], [2, 0], [0, 2]], float)      # self-crossing bow tie
    assert not opoly.polygon_metrics(bow)["valid"]
    touch = np.array([[0, 0], [0, 2], [2, 2], [2, 4], [4, 4], [4, 2], [2, 2], [2, 0]], float)   # figure 8 touching at (2,2)
    assert not opoly.polygon_metrics(touch)["valid"]
    spike = np.array([[0, 0], [0, 2], [2, 2], [4, 2], [2, 2], [2, 0]], float)   # out-and-back whisker
    assert not opoly.polygon_metrics(spike)["valid"]
    line = np.array([[0, 0], [1, 0], [2, 0], [3, 0]], float)
    assert not opoly.polygon_metrics(line)["valid"]


# ---- product host polygoniser vs the oracle ----------------------------------------------------------------
def _compare_with_oracle(m, cm, scale, origin, cells, xy):
    ref = opoly.post_process_tile(m, cm, origin, scale)
    valid = cells[cells["valid"] == 1]
    assert len(valid) == len(ref)
    labs = [l for l in np.unique(m) if l]
    for c, lab in zip(cells, labs):
        # every contour (valid or not) equals OpenCV-order contours[0]
        ys, xs = np.nonzero(m == lab)
        cont = opoly.find_contours_external_simple((m == lab)[ys.min(): ys.max() + 1, xs.min(): xs.max() + 1])[0]
        exp = (cont + [xs.min(), ys.min()]) * scale + np.asarray(origin)
        assert np.array_equal(xy[c["offset"]: c["offset"] + c["n_pts"]], exp), lab
    for c, r in zip(valid, ref):
        assert c["area"] == r["area"] and c["perimeter"] == pytest.approx(r["perimeter"], rel=1e-14)
        assert abs(c["cx"] - r["centroid_raw"][0]) < 1e-9 and abs(c["cy"] - r["centroid_raw"][1]) < 1e-9
        assert c["cls"] - 1 == r["class_int"]
    return len(ref)


def test_host_polygonizer_equals_oracle_hand_cases():
    m = np.zeros((48, 64), np.uint16)
    m[5:15, 10:30] = 1
    yy, xx = np.mgrid[:48, :64]
    m[(yy - 30) ** 2 + (xx - 30) ** 2 <= 64] = 2
    m[2, 2] = 3                                              # single pixel -> invalid (< 4 points)
    m[20:22, 44:52] = 4; m[40:43, 44:52] = 4                 # two components: the last one counts
    m[18, 2:9] = 5                                           # line -> invalid
    m[24:27, 2:5] = 6; m[26:29, 4:7] = 6                     # two squares sharing a corner pixel
    m[34:40, 8:10] = 7; m[34:40, 14:16] = 7; m[38:40, 8:16] = 7   # U shape
    m[10:14, 40:50] = 8; m[11, 50:56] = 8                    # blob with a 1-px whisker (spike -> invalid ring)
    recs = _records(m)
    cm = np.zeros(m.shape, np.uint8)
    for r in recs:
        cm[m == r["label"]] = r["cls"]
    for scale, origin in ((1.0, (0, 0)), (2.0, (100, 200)), (2.2727, (1017.0, 338.0))):
        cells, xy = postprocess.polygonize_tile(m, recs, scale, origin)
        n = _compare_with_oracle(m, cm, scale, origin, cells, xy)
        assert n == 5 and cells["valid"].tolist() == [1, 1, 0, 1, 0, 1, 1, 0]


@pytest.mark.parametrize("seed,thr", [(0, 0.02), (1, 0.0), (2, 0.05)])
def test_host_polygonizer_equals_oracle_random_blobs(seed, thr):
    rng = np.random.default_rng(seed)
    from scipy.ndimage import binary_fill_holes, gaussian_filter, label
    img = gaussian_filter(rng.standard_normal((96, 128)), 2.5) > thr
    lab, n = label(binary_fill_holes(img))
    # merge some labels so that multi-component instances occur
    lab[lab == n] = 1
    m = lab.astype(np.uint16)
    recs = _records(m)
    cm = np.zeros(m.shape, np.uint8)
    for r in recs:
        cm[m == r["label"]] = r["cls"]
    cells, xy = postprocess.polygonize_tile(m, recs, 1.0, (7, 9))
    assert _compare_with_oracle(m, cm, 1.0, (7, 9), cells, xy) > 5


# ---- the reference's own PostProcessor loop (tests/golden/make_golden_postprocessor.py) ---------------------------------
def _pp_case(g, ev, k):
    case, cx, cy, labelled = (int(v) for v in g[f"pp_{k}_cfg"])
    masks = ev[f"ev_{case}_masks"].astype(np.uint16)
    cm = ev[f"ev_{case}_class_masks"].astype(np.uint8)
    if int(g[f"pp_{k}_extra"]):
        masks = masks.copy(); cm = cm.copy()
        n = int(masks.max())
        masks[0, 0] = n + 1; masks[5, 250:252] = n + 2
        cm[0, 0] = 1; cm[5, 250:252] = 2
    return masks, cm, (cx, cy), float(g[f"pp_{k}_scale"]), bool(labelled)


def test_oracle_and_host_polygonizer_equal_reference_postprocessor():
    """``PostProcessor.__call__`` (predict_wsi.py:578-656) run by the reference itself on the id / class maps its own
    ``eval`` produced: the oracle's restatement of the loop and the product's host polygoniser + cell dict give the same
    cells -- vertices, area, perimeter, rounded centroid, class, label, colour, dict keys, rejection count."""
    import json
    import os
    from classpose_amd import geojson
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, ev = np.load(os.path.join(gd, "reference_postprocessor.npz")), np.load(os.path.join(gd, "reference_eval.npz"))
    labels = ["Alpha", "Beta", "Gamma"]
    for k in range(int(g["pp_n"])):
        masks, cm, coords, scale, labelled = _pp_case(g, ev, k)
        offs = np.concatenate([[0], np.cumsum(g[f"pp_{k}_n_pts"])])
        want_xy = [g[f"pp_{k}_coords"][offs[i]: offs[i + 1]] for i in range(len(offs) - 1)]
        # the oracle's loop
        ref = opoly.post_process_tile(masks, cm if labelled else None, coords, scale)
        assert len(ref) == len(want_xy)
        assert int(g[f"pp_{k}_n_invalid"]) == len(np.unique(masks)) - 1 - len(ref)
        for r, xy, a, p, c, ci in zip(ref, want_xy, g[f"pp_{k}_area"], g[f"pp_{k}_perimeter"], g[f"pp_{k}_centroid"], g[f"pp_{k}_class_int"]):
            assert np.array_equal(r["coords"], xy[:-1]) and np.array_equal(xy[-1], xy[0])      # the reference closes the ring
            assert r["area"] == a and r["perimeter"] == p and r["centroid"] == c.tolist()
            assert r["class_int"] == (ci if labelled else 0)
        # the product's host polygoniser + cell dict
        recs = _records(masks, cm)
        cells, pool = postprocess.polygonize_tile(masks, recs, scale, coords)
        valid = cells[cells["valid"] == 1]
        assert len(valid) == len(want_xy)
        names, colors = json.loads(str(g[f"pp_{k}_label"])), g[f"pp_{k}_color"]
        for i, c in enumerate(valid):
            ring = pool[c["offset"]: c["offset"] + c["n_pts"]]
            assert np.array_equal(ring, want_xy[i][:-1])
            assert c["area"] == g[f"pp_{k}_area"][i] and c["perimeter"] == pytest.approx(g[f"pp_{k}_perimeter"][i], rel=1e-14)
            d = geojson.cell_dict(ring.tolist(), int(c["cls"]) if labelled else 0, labels if labelled else None, c["area"],
                                  c["perimeter"], np.round([c["cx"], c["cy"]], 2).tolist())
            assert sorted(d.keys()) == json.loads(str(g[f"pp_{k}_keys"]))
            assert d["coords"] == want_xy[i].tolist() and d["centroid"] == g[f"pp_{k}_centroid"][i].tolist()
            assert d["label"] == names[i] and d["color"] == colors[i].tolist() and d["class_int"] == int(g[f"pp_{k}_class_int"][i])
