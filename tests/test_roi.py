"""Host ROI geometry (classpose_amd.roi): predicates, ROI tile grid, cell filter.

The reference computes these with shapely (absent here): the expectations below are
hand-derived from the definitions of intersects / within / intersection length."""
import json

import numpy as np
import pytest

from classpose_amd import roi


def _square(x0, y0, s):
    return [[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s], [x0, y0]]


L_SHAPE = [[0, 0], [100, 0], [100, 40], [40, 40], [40, 100], [0, 100]]


def test_locate_and_within():
    p = roi.RoiPolygon(_square(10, 10, 80), holes=[_square(40, 40, 20)])
    assert p.locate(20, 20) == 1 and p.locate(50, 50) == -1 and p.locate(5, 50) == -1
    assert p.locate(10, 30) == 0 and p.locate(40, 45) == 0 and p.locate(90, 90) == 0
    assert p.contains_point_strict(39.9, 50) and not p.contains_point_strict(40, 50)
    assert abs(p.area - (6400 - 400)) < 1e-9
    q = roi.RoiPolygon(L_SHAPE)
    assert q.locate(20, 80) == 1 and q.locate(80, 80) == -1 and q.locate(40, 40) == 0 and q.locate(70, 40) == 0


def test_intersects_square_closed_semantics():
    p = roi.RoiPolygon(_square(10, 10, 80), holes=[_square(40, 40, 20)])
    assert p.intersects_square(0, 0, 10)            # touches at the corner (10, 10): closed sets intersect
    assert not p.intersects_square(0, 0, 9.99)
    assert p.intersects_square(-50, -50, 500)       # square contains the polygon
    assert p.intersects_square(20, 20, 5)           # square inside the polygon
    assert not p.intersects_square(45, 45, 10)      # square strictly inside the hole
    assert p.intersects_square(45, 45, 15)          # reaches the hole's boundary
    assert p.intersects_square(85, 50, 30) and not p.intersects_square(91, 50, 30)
    q = roi.RoiPolygon(L_SHAPE)
    assert not q.intersects_square(50, 50, 40) and q.intersects_square(30, 30, 40)


def test_chords_and_maximum_lengths():
    q = roi.RoiPolygon(L_SHAPE)
    assert q.chord_lengths(20.0, 0) == [100.0] and q.chord_lengths(70.0, 0) == [40.0]
    assert q.chord_lengths(0.0, 0) == [100.0]               # collinear boundary edge counts (closed polygon)
    assert q.chord_lengths(100.0, 0) == [40.0]
    p = roi.RoiPolygon(_square(0, 0, 90), holes=[_square(30, 30, 30)])
    assert p.chord_lengths(45.0, 1) == [30.0, 30.0]          # split by the hole
    assert roi.get_maximum_lengths(q) == (100.0, 100.0)
    r = roi.RoiPolygon([[0, 0], [600, 0], [600, 300], [0, 300]])
    assert roi.get_maximum_lengths(r) == (300.0, 600.0)


def test_invalid_ring_raises():
    with pytest.raises(ValueError):
        roi.RoiPolygon([[0, 0], [10, 10], [10, 0], [0, 10]])    # bow tie
    with pytest.raises(ValueError):
        roi.RoiPolygon([[0, 0], [1, 1]])


def test_get_coords_roi_hand_derived():
    """600 x 300 ROI at (1000, 2000), ts=1, read tile 512 / overlap 64, model overlap 64:
    box padded by 32 -> x in [968, 1632), y in [1968, 2332); narrowest max chord 300 ->
    cts = 300; stride 236; last row/column pulled back to cmax - cts"""
    r = roi.RoiPolygon([[1000, 2000], [1600, 2000], [1600, 2300], [1000, 2300]])
    got = roi.get_coords_roi([r], 512, 64, 64, 1.0)
    xs = [968, 968 + 236, 1632 - 300]
    ys = [1968, 2332 - 300]
    assert got == [((x, y), 300) for x in xs for y in ys]
    assert all(roi.check_tile_in_cnts(c, s, 1.0, [r]) for c, s in got)
    # tiny ROI -> MIN_TILE_SIZE; huge ROI -> tile_size; downsampled level scales coordinates
    small = roi.RoiPolygon(_square(500, 500, 40))
    (c0, s0), = roi.get_coords_roi([small], 512, 64, 64, 1.0)[:1]
    assert s0 == 256 and c0 == (572 - 256, 572 - 256)      # box 468..572 is narrower than the tile: pulled back
    big = roi.RoiPolygon(_square(0, 0, 4000))
    got = roi.get_coords_roi([big], 512, 64, 64, 2.0)
    assert got[0] == ((-64, -64), 512) and got[1][0] == (-64, int((-32 + 448) * 2.0))


def test_load_roi_polygons_and_filter(tmp_path):
    fc = {"type": "FeatureCollection", "features": [
        {"type": "Feature", "geometry": {"type": "Polygon", "coordinates": [_square(0, 0, 100)]},
         "properties": {"classification": {"name": "tumour"}}},
        {"type": "Feature", "geometry": {"type": "MultiPolygon",
                                         "coordinates": [[_square(50, 50, 100)], [_square(300, 300, 10)]]},
         "properties": {}},
        {"type": "Feature", "geometry": {"type": "LineString", "coordinates": [[400, 400], [420, 400], [410, 420]]},
         "properties": {}},
        {"type": "Feature", "geometry": None, "properties": {}}]}
    path = tmp_path / "roi.geojson"
    path.write_text(json.dumps(fc))
    polys, by_class = roi.load_roi_polygons(str(path), group_by_class=True)
    assert len(polys) == 4 and sorted(by_class) == ["tumour", "unknown"] and len(by_class["unknown"]) == 3
    path2 = tmp_path / "single.geojson"
    path2.write_text(json.dumps(fc["features"][0]))
    assert len(roi.load_roi_polygons(str(path2))) == 1
    empty = tmp_path / "empty.geojson"
    empty.write_text(json.dumps({"type": "FeatureCollection", "features": []}))
    assert roi.load_roi_polygons(str(empty)) is None

    def cell(x, y):
        return {"properties": {"measurements": [{"name": "area", "value": 1.0}, {"name": "centroidX", "value": x},
                                                {"name": "centroidY", "value": y}]}}
    cells = [cell(10, 10), cell(75, 75), cell(200, 200), cell(100, 20), cell(305, 305), cell(410, 405)]
    kept = roi.filter_cells_by_contours(cells, polys)
    # (75,75) lies in two overlapping ROIs -> returned twice, like STRtree.query(..., "within");
    # (100,20) is on a boundary -> not within; (200,200) is outside everything
    assert [roi.get_cell_centroid(c) for c in kept] == [[10, 10], [75, 75], [75, 75], [305, 305], [410, 405]]
    assert roi.filter_cells_by_contours(cells, []) == cells


def test_vectorised_predicates_match_scalar():
    """numpy paths (long raster rings, many points) == the scalar definitions, incl. boundary points"""
    from scipy import ndimage
    from classpose_amd import grandqc
    rng = np.random.default_rng(5)
    m = (ndimage.gaussian_filter(rng.standard_normal((60, 80)), 3.0) > 0).astype(np.uint8)
    cnts, parent = grandqc.find_contours_ccomp(m)
    outer = int(np.argmax([len(c) if p == -1 else 0 for c, p in zip(cnts, parent)]))
    poly = roi.RoiPolygon(cnts[outer] * 3.0, [cnts[j] * 3.0 for j in np.nonzero(parent == outer)[0]], validate=False)
    assert len(poly.exterior) > 64
    xs = np.concatenate([rng.uniform(-5, 245, 3000), rng.integers(0, 80, 500) * 3.0])   # random + on-lattice (boundary) points
    ys = np.concatenate([rng.uniform(-5, 185, 3000), rng.integers(0, 60, 500) * 3.0])
    got = poly.contains_points_strict(xs, ys)

    def scalar_locate(p, x, y):
        inside = False
        for ring in p.rings:
            ins = False
            for k in range(len(ring) - 1):
                a, b = ring[k], ring[k + 1]
                if roi._orient(a, b, (x, y)) == 0 and roi._on_segment(a, b, (x, y)):
                    return 0
                if (a[1] > y) != (b[1] > y) and a[0] + (y - a[1]) * (b[0] - a[0]) / (b[1] - a[1]) > x:
                    ins = not ins
            inside ^= ins
        return 1 if inside else -1
    want = np.array([scalar_locate(poly, x, y) > 0 for x, y in zip(xs, ys)])
    assert np.array_equal(got, want) and want.any() and not want.all()
    assert all(poly.locate(x, y) == scalar_locate(poly, x, y) for x, y in zip(xs[::7], ys[::7]))
    for _ in range(300):
        x0, y0, s = rng.uniform(-20, 240), rng.uniform(-20, 180), rng.choice([2.0, 9.0, 40.0])
        sq = [[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s], [x0, y0]]
        brute = any(roi._segments_intersect(r[k], r[k + 1], sq[j], sq[j + 1])
                    for r in poly.rings for k in range(len(r) - 1) for j in range(4)) \
            or scalar_locate(poly, x0, y0) >= 0 or (x0 <= poly.exterior[0][0] <= x0 + s and y0 <= poly.exterior[0][1] <= y0 + s)
        assert poly.intersects_square(x0, y0, s) == brute


def test_intersection_area_known_answers_and_monte_carlo():
    """roi.intersection_area = shapely's a.intersection(b).area (predict_wsi.py:1822-1826), by an exact y-sweep"""
    from classpose_amd.roi import RoiPolygon, intersection_area
    sq = lambda x0, y0, x1, y1: [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]
    a = RoiPolygon(sq(0, 0, 10, 10))
    assert intersection_area(a, RoiPolygon(sq(5, 5, 15, 15))) == 25.0
    assert intersection_area(a, RoiPolygon(sq(10, 0, 20, 10))) == 0.0                # touching edge
    assert intersection_area(a, RoiPolygon(sq(20, 20, 30, 30))) == 0.0                # disjoint
    assert intersection_area(a, a) == 100.0
    assert intersection_area(a, RoiPolygon(sq(2, 2, 4, 4))) == 4.0                    # contained
    holed = RoiPolygon(sq(0, 0, 10, 10), [sq(3, 3, 7, 7)])
    assert intersection_area(holed, RoiPolygon(sq(0, 0, 10, 10))) == 84.0
    assert intersection_area(holed, RoiPolygon(sq(4, 4, 6, 6))) == 0.0               # inside the hole
    assert intersection_area(holed, RoiPolygon(sq(2, 2, 8, 8))) == 36.0 - 16.0
    tri = RoiPolygon([[0, 0], [10, 0], [0, 10]])
    assert abs(intersection_area(tri, RoiPolygon(sq(0, 0, 5, 5))) - 25.0) < 1e-12
    assert abs(intersection_area(tri, RoiPolygon(sq(4, 4, 10, 10))) - 2.0) < 1e-12    # corner triangle (4,4),(6,4),(4,6)
    # symmetric, and agrees with a dense point-sampling estimate on ragged polygons
    rng = np.random.default_rng(3)

    def blob(cx, cy, r, n):
        th = np.sort(rng.uniform(0, 2 * np.pi, n))
        rr = r * rng.uniform(0.5, 1.0, n)
        return RoiPolygon(np.stack([cx + rr * np.cos(th), cy + rr * np.sin(th)], 1))
    for k in range(4):
        p, q = blob(50, 50, 40, 23), blob(65 + 5 * k, 55, 35, 31)
        ia = intersection_area(p, q)
        assert abs(ia - intersection_area(q, p)) < 1e-9 * max(ia, 1.0)
        xs, ys = rng.uniform(0, 120, 400_000), rng.uniform(0, 120, 400_000)
        mc = float((p.contains_points_strict(xs, ys) & q.contains_points_strict(xs, ys)).mean()) * 120 * 120
        assert abs(ia - mc) < 0.02 * max(mc, 1.0), (ia, mc)
        assert ia <= min(p.area, q.area) + 1e-9


def test_make_valid_splits_self_intersecting_rings(tmp_path):
    """predict_wsi.make_valid (:1050): a bow tie is repaired into its two triangles instead of raising"""
    import json
    from classpose_amd import roi
    bow = [[0, 0], [10, 10], [10, 0], [0, 10]]
    parts = roi.make_valid_polygon(bow)
    assert len(parts) == 2 and sorted(round(p.area, 9) for p in parts) == [25.0, 25.0]
    assert any(p.contains_point_strict(2, 5) for p in parts) and any(p.contains_point_strict(8, 5) for p in parts)
    assert not any(p.contains_point_strict(5, 2) for p in parts)
    sq = roi.make_valid_polygon([[0, 0], [4, 0], [4, 4], [0, 4]], [[[1, 1], [2, 1], [2, 2], [1, 2]]])
    assert len(sq) == 1 and sq[0].area == 15.0
    # triple bow tie (four lobes: two triangles, two diamonds) and a hole that lands in the first lobe
    z = [[0, 0], [4, 4], [8, 0], [12, 4], [12, 0], [8, 4], [4, 0], [0, 4]]
    parts = roi.make_valid_polygon(z, [[[0.5, 1.5], [1.0, 1.5], [1.0, 2.5], [0.5, 2.5]]])
    assert len(parts) == 4 and abs(sum(p.area for p in parts) - (4 + 8 + 8 + 4 - 0.5)) < 1e-9
    assert sum(len(p.holes) for p in parts) == 1
    f = tmp_path / "roi.geojson"
    f.write_text(json.dumps({"type": "FeatureCollection", "features": [
        {"type": "Feature", "geometry": {"type": "Polygon", "coordinates": [bow + [bow[0]]]},
         "properties": {"classification": {"name": "x"}}}]}))
    polys, by_class = roi.load_roi_polygons(str(f), group_by_class=True)
    assert len(polys) == 2 and len(by_class["x"]) == 2
