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


def test_streaming_writer_equals_dict_pipeline(tmp_path):
    """the CLI's struct-of-arrays path (dedup on rounded centroids, bounds offset, streamed JSON) produces
    the same FeatureCollections as the reference-shaped list-of-dict functions + json.dump"""
    import json
    from classpose_amd.entrypoints.predict_wsi import CELL_ROW
    rng = np.random.default_rng(4)
    n = 400
    labels = ["a", "b", "c"]
    cells = np.zeros(n, CELL_ROW)
    cells["n_pts"] = rng.integers(4, 9, n)
    cells["cls"] = rng.integers(0, 4, n)                    # includes class 0 -> labels[-1]
    cells["area"] = rng.uniform(20, 300, n)
    cells["perimeter"] = rng.uniform(10, 80, n)
    cells["cx"] = rng.uniform(0, 300, n)
    cells["cy"] = rng.uniform(0, 300, n)
    cells["cx"][50:80] = cells["cx"][20:50] + rng.uniform(-3, 3, 30)      # near-duplicates
    cells["cy"][50:80] = cells["cy"][20:50] + rng.uniform(-3, 3, 30)
    xy = rng.uniform(0, 300, (int(cells["n_pts"].sum()), 2)) * 0.37
    offs = np.concatenate([[0], np.cumsum(cells["n_pts"])])
    bounds = (12.5, 7.25)
    # dict pipeline (what the reference does)
    feats = []
    for i, c in enumerate(cells):
        centroid = np.round([c["cx"], c["cy"]], 2).tolist()
        feats.append(geojson.to_geojson_polygon(geojson.cell_dict(xy[offs[i]:offs[i + 1]].tolist(), int(c["cls"]), labels,
                                                                  c["area"], c["perimeter"], centroid)))
    feats = geojson.deduplicate(feats)
    feats = [geojson.apply_bounds_offset_to_feature(f, *bounds) for f in feats]
    ref_cont = json.loads(json.dumps({"type": "FeatureCollection", "features": feats}))
    ref_cent = json.loads(json.dumps({"type": "FeatureCollection", "features": geojson.polygons_to_centroids(feats)}))
    # array pipeline
    keep = geojson.dedup_indices(geojson.rounded_centroids(cells), cells["area"])
    assert 0 < len(keep) < n
    n_written = geojson.write_feature_collections(tmp_path / "c.json", tmp_path / "p.json", cells, xy, keep, labels, bounds)
    got_cont, got_cent = json.load(open(tmp_path / "c.json")), json.load(open(tmp_path / "p.json"))
    assert n_written == len(ref_cont["features"]) == len(got_cont["features"]) == len(got_cent["features"])

    def strip(fc):
        return [{k: v for k, v in f.items() if k != "id"} for f in fc["features"]]
    assert strip(got_cont) == strip(ref_cont) and strip(got_cent) == strip(ref_cent)
    assert len({f["id"] for f in got_cont["features"]}) == n_written
    # text level: same separators as json.dump
    txt = open(tmp_path / "c.json").read()
    assert txt.startswith('{"type": "FeatureCollection", "features": [{"type": "Feature", "id": "')
    # class-less labels and an empty selection
    geojson.write_feature_collections(tmp_path / "e.json", tmp_path / "e2.json", cells, xy, [], None)
    assert json.load(open(tmp_path / "e.json")) == {"type": "FeatureCollection", "features": []}
    geojson.write_feature_collections(tmp_path / "n.json", tmp_path / "n2.json", cells, xy, [3, 3], None)
    f = json.load(open(tmp_path / "n.json"))["features"]
    assert len(f) == 2 and f[0]["properties"]["classification"] == {"name": "cell", "color": [0, 168, 132]}
    # vertex gather
    sel = np.array([5, 2, 9])
    g = geojson.gather_vertices(xy, offs[sel], cells["n_pts"][sel])
    assert np.array_equal(g, np.concatenate([xy[offs[i]:offs[i + 1]] for i in sel]))


def test_native_writer_is_bytewise_json_dump(tmp_path):
    """cpx_write_geojson against json.dumps of the reference-shaped feature dicts, byte for byte (ids masked),
    on values that exercise every branch of float.__repr__: integers, exponents on both sides of the
    fixed / scientific switch (1e-4, 1e16), 17-digit values, negative zero, NaN / +-Infinity (json.dump's spelling, in
    the measurements, the Point coordinates and the ring alike), and a bounds offset."""
    import json
    import re
    from classpose_amd.entrypoints.predict_wsi import CELL_ROW
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 1.0, 1e15, 1e16, 1.5e16, 1e-4, 9.999e-5, 1e-5, 123456789012345.6, 2.0 / 3, 0.1 + 0.2,
                        1e22, 5e-324, 1.7976931348623157e308, 4.35, 100.0, 65536.5, 1234567.891, -3.25, 1e-7, 12345678.9,
                        np.nan, np.inf, -np.inf])
    n = 64
    cells = np.zeros(n, CELL_ROW)
    cells["n_pts"] = rng.integers(3, 7, n)
    cells["cls"] = rng.integers(0, 3, n)
    cells["area"] = np.resize(special, n)
    cells["perimeter"] = np.resize(special[::-1], n)
    cells["cx"] = rng.uniform(0, 1e5, n)
    cells["cy"] = rng.uniform(0, 1e5, n)
    cells["cx"][:3] = [np.nan, np.inf, -np.inf]                                 # non-finite centroids -> Point coordinates too
    tot = int(cells["n_pts"].sum())
    xy = np.stack([np.resize(special, tot), rng.uniform(-10, 1e5, tot)], 1)
    xy[5:40] = rng.integers(0, 40000, (35, 2)).astype(np.float64) * 0.5        # what tile contours look like
    offs = np.concatenate([[0], np.cumsum(cells["n_pts"])])
    labels = ['a "quoted" name', "b", "ünï"]                                    # json.dumps escapes both
    keep = [int(k) for k in rng.permutation(n)[:50]] + [3, 3, 0, 1, 2]
    for bounds in [(0.0, 0.0), (12.5, -7.25)]:
        feats = []
        for i in keep:
            c = cells[i]
            centroid = np.round([c["cx"], c["cy"]], 2).tolist()
            f = geojson.to_geojson_polygon(geojson.cell_dict(xy[offs[i]:offs[i + 1]].tolist(), int(c["cls"]), labels,
                                                            c["area"], c["perimeter"], centroid))
            feats.append(geojson.apply_bounds_offset_to_feature(f, *bounds) if bounds != (0.0, 0.0) else f)
        want_c = json.dumps({"type": "FeatureCollection", "features": feats})
        want_p = json.dumps({"type": "FeatureCollection", "features": geojson.polygons_to_centroids(feats)})
        geojson.write_feature_collections(tmp_path / "c.json", tmp_path / "p.json", cells, xy, keep, labels, bounds, n_threads=3)
        mask = lambda t: re.sub(r'"id": "[0-9a-f]{8}-[0-9a-f]{4}-4[0-9a-f]{3}-[89ab][0-9a-f]{3}-[0-9a-f]{12}"', '"id": "X"', t)
        got_c, got_p = open(tmp_path / "c.json").read(), open(tmp_path / "p.json").read()
        assert mask(got_c) == mask(want_c)
        assert mask(got_p) == mask(want_p)
        assert '"id": "X"' in mask(got_c) and mask(got_c).count('"id": "X"') == len(keep)
        ids = re.findall(r'"id": "([0-9a-f-]{36})"', got_c + got_p)
        assert len(set(ids)) == 2 * len(keep)                                    # every feature has its own uuid4


def test_native_writer_many_chunks(tmp_path):
    """more cells than one chunk and more chunks than the in-memory window: order and separators hold"""
    import json
    from classpose_amd.entrypoints.predict_wsi import CELL_ROW
    n = 8192 * 9 + 17
    cells = np.zeros(n, CELL_ROW)
    cells["n_pts"] = 3
    cells["cls"] = 1
    cells["area"] = np.arange(n)
    cells["cx"] = np.arange(n) * 0.25
    xy = np.zeros((3 * n, 2))
    xy[:, 0] = np.arange(3 * n)
    geojson.write_feature_collections(tmp_path / "c.json", tmp_path / "p.json", cells, xy, np.arange(n), None, n_threads=2)
    c = json.load(open(tmp_path / "c.json"))["features"]
    p = json.load(open(tmp_path / "p.json"))["features"]
    assert len(c) == len(p) == n
    assert [f["properties"]["measurements"][0]["value"] for f in c] == list(map(float, range(n)))
    assert [f["geometry"]["coordinates"][0] for f in p] == [i * 0.25 for i in range(n)]
    assert c[-1]["geometry"]["coordinates"][0][0] == [3.0 * (n - 1), 0.0] == c[-1]["geometry"]["coordinates"][0][3]
    with pytest.raises(Exception):
        geojson.write_feature_collections(tmp_path / "c.json", tmp_path / "p.json", cells, xy, [n], None)   # index out of range
