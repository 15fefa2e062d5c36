"""CPU: host polygonisation (cpx_polygonize_host) and the GeoJSON / dedup layer."""
import json

import numpy as np
import pytest

from classpose_amd import geojson, postprocess
from classpose_amd.engine import RECORD_DTYPE


def _records(m):
    labs = [l for l in np.unique(m) if l]
    recs = np.zeros(len(labs), RECORD_DTYPE)
    for i, l in enumerate(labs):
        ys, xs = np.nonzero(m == l)
        recs[i] = (0, l, int(l) % 7, len(ys), ys.min(), xs.min(), ys.max() + 1, xs.max() + 1, ys.sum(), xs.sum())
    return recs


def test_rectangle_disc_single_pixel_and_two_components():
    m = np.zeros((40, 50), np.uint16)
    m[5:15, 10:30] = 1
    yy, xx = np.mgrid[:40, :50]
    m[(yy - 28) ** 2 + (xx - 30) ** 2 <= 64] = 2
    m[2, 2] = 3
    m[20:22, 40:48] = 4
    m[30:32, 40:48] = 4
    cells, xy = postprocess.polygonize_tile(m, _records(m), 2.0, (100, 200))
    r = xy[cells[0]["offset"]: cells[0]["offset"] + cells[0]["n_pts"]]
    # OpenCV order for a filled rectangle: top-left, down, right, up (pixel-centre coordinates)
    assert r.tolist() == [[120.0, 210.0], [120.0, 228.0], [158.0, 228.0], [158.0, 210.0]]
    assert cells[0]["area"] == 19 * 9 * 4 and cells[0]["perimeter"] == 2 * (38 + 18)
    assert (cells[0]["cx"], cells[0]["cy"]) == (139.0, 219.0) and cells[0]["valid"] == 1
    # disc: centroid at the centre, area a bit below pi r^2 (contour through pixel centres)
    assert abs(cells[1]["cx"] - 160) < 1e-9 and abs(cells[1]["cy"] - 256) < 1e-9
    assert 0.8 * np.pi * 64 * 4 < cells[1]["area"] < np.pi * 64 * 4
    assert cells[2]["n_pts"] == 1 and cells[2]["valid"] == 0                 # < 4 points -> dropped
    assert cells[3]["valid"] == 1 and xy[cells[3]["offset"]][1] == 260.0     # last-found component


def test_thin_and_pinched_shapes_are_invalid():
    m = np.zeros((20, 20), np.uint16)
    m[3, 2:10] = 1                     # 1-px line: 2 contour points
    m[8:11, 2:5] = 2; m[10:13, 4:7] = 2   # two squares sharing one corner pixel -> ring touches itself
    cells, xy = postprocess.polygonize_tile(m, _records(m), 1.0, (0, 0))
    assert cells[0]["valid"] == 0
    assert cells[1]["n_pts"] >= 4


def test_random_blobs_area_matches_shoelace_of_pixel_centres():
    rng = np.random.default_rng(0)
    from scipy.ndimage import binary_fill_holes, gaussian_filter, label
    img = gaussian_filter(rng.standard_normal((120, 160)), 4) > 0.02
    lab, n = label(binary_fill_holes(img))
    m = lab.astype(np.uint16)
    recs = _records(m)
    cells, xy = postprocess.polygonize_tile(m, recs, 1.0, (0, 0))
    assert len(cells) == n
    for c, r in zip(cells, recs):
        if not c["valid"]:
            continue
        # contour through boundary-pixel centres: enclosed area is within the pixel count and
        # pixel count minus ~half the perimeter (Pick-like bound)
        assert c["area"] <= r["area"]
        assert c["area"] >= r["area"] - c["perimeter"] / 2 - 2
        assert r["x0"] <= c["cx"] <= r["x1"] and r["y0"] <= c["cy"] <= r["y1"]


def test_geojson_schema_and_dedup_golden(golden):
    _, js = golden
    g = js["geojson"]
    pts = np.array(g["points"])
    kept = geojson.dedup_indices(pts[:, :2].tolist(), pts[:, 2].tolist())
    assert kept == g["kept_ids"]
    cell = dict(id="0", coords=g["feature0"]["geometry"]["coordinates"][0],
                label="a", color=[1, 2, 3], area=pts[0, 2], perimeter=4.0, centroid=[pts[0, 0], pts[0, 1]])
    assert geojson.to_geojson_polygon(cell) == g["feature0"]
    cen = geojson.polygons_to_centroids([g["feature0"]])[0]
    cen["id"] = "uuid"
    assert cen == g["centroid0"]
    shifted = geojson.apply_bounds_offset_to_feature(json.loads(json.dumps(g["feature1"])), 10.0, 20.0)
    assert shifted == g["feature1_shifted"]
    feats = [geojson.to_geojson_polygon(dict(id=str(i), coords=[[x, y]], label="a", color=[0, 0, 0],
                                             area=a, perimeter=1.0, centroid=[x, y]))
             for i, (x, y, a) in enumerate(g["points"])]
    assert [int(f["id"]) for f in geojson.deduplicate(feats)] == g["kept_ids"]


def test_cell_dict_class_zero_indexes_last_label():
    d = geojson.cell_dict([[0, 0], [1, 0], [1, 1], [0, 1]], 0, ["A", "B", "C"], 1.0, 4.0, [0.5, 0.5])
    assert d["label"] == "C" and d["color"] == geojson.COLORMAP[-1] and d["class_int"] == -1
    assert d["coords"][0] == d["coords"][-1] and len(d["coords"]) == 5
    d = geojson.cell_dict([[0, 0], [1, 0], [1, 1], [0, 1]], 2, ["A", "B", "C"], 1.0, 4.0, [0.5, 0.5])
    assert d["label"] == "B" and d["color"] == [255, 255, 179]
