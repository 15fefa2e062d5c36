"""ROI handling of the WSI path (host side): GeoJSON ROI loading, ROI-driven tile grid,
tile/ROI intersection test and the final centroid-in-ROI cell filter.

Mirrors ``load_roi_polygons`` (/root/reference/src/classpose/entrypoints/predict_wsi.py:1086-1161),
``get_maximum_lengths`` (:1164-1200), ``SlideLoader._get_coords_roi`` (:324-364),
``_check_tile_in_cnts`` (:406-417), ``_align_roi_tree_to_slide_bounds`` (:280-304) and
``filter_cells_by_contours`` (:1247-1278).  The reference delegates the geometry to
shapely/GEOS (absent from this image); the predicates it uses -- ``intersects`` between a
polygon and an axis-aligned square, point ``within`` polygon, chord lengths of axis-parallel
lines -- are restated here on float64 with the same open/closed conventions (``intersects`` is
closed, ``within`` excludes the boundary).  ``make_valid``: a self-intersecting ROI ring is cut into its
simple loops (``make_valid_polygon``), which is what GEOS' structure-preserving repair yields wherever the
loops do not overlap each other.
"""
from __future__ import annotations

import json

import numpy as np

MIN_TILE_SIZE = 256


class RoiPolygon:
    """One polygon: closed exterior ring + closed hole rings, float64 (n, 2) arrays."""

    def __init__(self, exterior, holes=(), validate: bool = True):
        """validate=False accepts rings that touch or retrace themselves (border-following contours of
        raster masks do): the predicates below are even-odd point-set tests and stay well defined;
        the reference repairs such rings with GEOS make_valid / buffer(0), which differs on
        measure-zero spikes only."""
        self.exterior = self._close(np.asarray(exterior, dtype=np.float64)[:, :2])
        self.holes = [self._close(np.asarray(h, dtype=np.float64)[:, :2]) for h in holes]
        self.holes = [h for h in self.holes if len(h) >= 4]
        for ring in self.rings:
            if len(ring) < 4:
                raise ValueError("ROI ring needs at least three distinct vertices")
            if validate and not _ring_is_simple(ring):
                raise ValueError("ROI ring self-intersects; repair it (e.g. shapely.make_valid) "
                                 "before passing it: GEOS' repair is not restated here")

    @staticmethod
    def _close(r: np.ndarray) -> np.ndarray:
        return r if (len(r) and np.array_equal(r[0], r[-1])) else np.concatenate([r, r[:1]])

    @property
    def rings(self):
        return [self.exterior] + self.holes

    @property
    def bounds(self):
        return (self.exterior[:, 0].min(), self.exterior[:, 1].min(),
                self.exterior[:, 0].max(), self.exterior[:, 1].max())

    @property
    def area(self) -> float:
        return abs(_ring_area(self.exterior)) - sum(abs(_ring_area(h)) for h in self.holes)

    @property
    def length(self) -> float:
        return float(sum(np.hypot(*np.diff(r, axis=0).T).sum() for r in self.rings))

    @property
    def centroid(self) -> tuple[float, float]:
        """area-weighted centroid (holes subtract), what shapely's Polygon.centroid returns"""
        a6 = cx = cy = 0.0
        for k, r in enumerate(self.rings):
            x, y = r[:-1, 0] - self.exterior[0, 0], r[:-1, 1] - self.exterior[0, 1]
            x1, y1 = r[1:, 0] - self.exterior[0, 0], r[1:, 1] - self.exterior[0, 1]
            cr = x * y1 - x1 * y
            sign = 1.0 if k == 0 else -1.0
            orient = 1.0 if cr.sum() >= 0 else -1.0
            a6 += sign * orient * cr.sum() * 3
            cx += sign * orient * ((x + x1) * cr).sum()
            cy += sign * orient * ((y + y1) * cr).sum()
        if a6 == 0:
            return float(self.exterior[0, 0]), float(self.exterior[0, 1])
        return float(self.exterior[0, 0] + cx / a6), float(self.exterior[0, 1] + cy / a6)

    def translate(self, xoff: float, yoff: float) -> "RoiPolygon":
        off = np.array([xoff, yoff])
        return RoiPolygon(self.exterior + off, [h + off for h in self.holes], validate=False)

    # -- predicates ----------------------------------------------------------------
    def locate(self, x: float, y: float) -> int:
        """+1 interior, 0 on the boundary, -1 exterior (holes are exterior)."""
        inside = False
        for ring in self.rings:
            loc = _locate_in_ring(ring, x, y)
            if loc == 0:
                return 0
            if loc > 0:
                inside = not inside
        return 1 if inside else -1

    def contains_point_strict(self, x: float, y: float) -> bool:
        """shapely ``Point.within(polygon)``: interior only."""
        return self.locate(x, y) > 0

    def _edge_index(self, nbins: int = 512):
        """edges of all rings bucketed by the y range they span (built once, for the vectorised tests)"""
        if getattr(self, "_eidx", None) is None:
            a = np.concatenate([r[:-1] for r in self.rings])
            b = np.concatenate([r[1:] for r in self.rings])
            y0, y1 = self.bounds[1], self.bounds[3]
            nb = int(max(1, min(nbins, len(a) // 4)))
            h = (y1 - y0) / nb if y1 > y0 else 1.0
            lo = np.clip(np.floor((np.minimum(a[:, 1], b[:, 1]) - y0) / h).astype(np.int64), 0, nb - 1)
            hi = np.clip(np.floor((np.maximum(a[:, 1], b[:, 1]) - y0) / h).astype(np.int64), 0, nb - 1)
            buckets = [[] for _ in range(nb)]
            for e in range(len(a)):
                for k in range(lo[e], hi[e] + 1):
                    buckets[k].append(e)
            self._eidx = (a, b, y0, h, nb, [np.asarray(k, dtype=np.int64) for k in buckets])
        return self._eidx

    def contains_points_strict(self, xs, ys) -> np.ndarray:
        """vectorised ``contains_point_strict`` (even-odd over all rings, boundary points excluded)"""
        xs, ys = np.asarray(xs, dtype=np.float64), np.asarray(ys, dtype=np.float64)
        out = np.zeros(len(xs), bool)
        bx0, by0, bx1, by1 = self.bounds
        cand = np.nonzero((xs >= bx0) & (xs <= bx1) & (ys >= by0) & (ys <= by1))[0]
        if len(cand) == 0:
            return out
        a, b, y0, h, nb, buckets = self._edge_index()
        bins = np.clip(np.floor((ys[cand] - y0) / h).astype(np.int64), 0, nb - 1)
        order = np.argsort(bins, kind="stable")
        cand, bins = cand[order], bins[order]
        starts = np.searchsorted(bins, np.arange(nb)); ends = np.searchsorted(bins, np.arange(nb), side="right")
        for k in range(nb):
            if starts[k] == ends[k] or len(buckets[k]) == 0:
                continue
            idx = cand[starts[k]:ends[k]]
            ea, eb = a[buckets[k]], b[buckets[k]]
            for s0 in range(0, len(idx), 8192):
                ii = idx[s0:s0 + 8192]
                px, py = xs[ii][:, None], ys[ii][:, None]
                ax, ay, bx, by = ea[None, :, 0], ea[None, :, 1], eb[None, :, 0], eb[None, :, 1]
                orient = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
                on = (orient == 0) & (px >= np.minimum(ax, bx)) & (px <= np.maximum(ax, bx)) & \
                     (py >= np.minimum(ay, by)) & (py <= np.maximum(ay, by))
                span = (ay > py) != (by > py)
                with np.errstate(divide="ignore", invalid="ignore"):
                    xi = ax + (py - ay) * (bx - ax) / (by - ay)
                cross = span & (xi > px)
                out[ii] = (cross.sum(1) % 2 == 1) & ~on.any(1)
        return out

    def intersects_square(self, x0: float, y0: float, size: float) -> bool:
        """shapely ``polygon.intersects(square)`` (closed sets share at least one point)."""
        x1, y1 = x0 + size, y0 + size
        bx0, by0, bx1, by1 = self.bounds
        if bx1 < x0 or bx0 > x1 or by1 < y0 or by0 > y1:
            return False
        if getattr(self, "_edges", None) is None:
            self._edges = (np.concatenate([r[:-1] for r in self.rings]), np.concatenate([r[1:] for r in self.rings]))
        a, b = self._edges
        near = ~((np.maximum(a[:, 0], b[:, 0]) < x0) | (np.minimum(a[:, 0], b[:, 0]) > x1) |
                 (np.maximum(a[:, 1], b[:, 1]) < y0) | (np.minimum(a[:, 1], b[:, 1]) > y1))
        if near.any():
            ea, eb = a[near], b[near]
            sq = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1], [x0, y0]])
            for j in range(4):
                c, d = sq[j], sq[j + 1]
                o1 = (eb[:, 0] - ea[:, 0]) * (c[1] - ea[:, 1]) - (eb[:, 1] - ea[:, 1]) * (c[0] - ea[:, 0])
                o2 = (eb[:, 0] - ea[:, 0]) * (d[1] - ea[:, 1]) - (eb[:, 1] - ea[:, 1]) * (d[0] - ea[:, 0])
                o3 = (d[0] - c[0]) * (ea[:, 1] - c[1]) - (d[1] - c[1]) * (ea[:, 0] - c[0])
                o4 = (d[0] - c[0]) * (eb[:, 1] - c[1]) - (d[1] - c[1]) * (eb[:, 0] - c[0])
                proper = ((o1 > 0) != (o2 > 0)) & ((o3 > 0) != (o4 > 0)) & (o1 != 0) & (o2 != 0) & (o3 != 0) & (o4 != 0)
                lo, hi = np.minimum(ea, eb), np.maximum(ea, eb)
                slo, shi = np.minimum(c, d), np.maximum(c, d)
                t1 = (o1 == 0) & (lo[:, 0] <= c[0]) & (c[0] <= hi[:, 0]) & (lo[:, 1] <= c[1]) & (c[1] <= hi[:, 1])
                t2 = (o2 == 0) & (lo[:, 0] <= d[0]) & (d[0] <= hi[:, 0]) & (lo[:, 1] <= d[1]) & (d[1] <= hi[:, 1])
                t3 = (o3 == 0) & (slo[0] <= ea[:, 0]) & (ea[:, 0] <= shi[0]) & (slo[1] <= ea[:, 1]) & (ea[:, 1] <= shi[1])
                t4 = (o4 == 0) & (slo[0] <= eb[:, 0]) & (eb[:, 0] <= shi[0]) & (slo[1] <= eb[:, 1]) & (eb[:, 1] <= shi[1])
                if (proper | t1 | t2 | t3 | t4).any():
                    return True
        # no boundary crossing: one contains the other, or they are disjoint
        if self.locate(x0, y0) >= 0:
            return True
        ex, ey = self.exterior[0]
        return bool(x0 <= ex <= x1 and y0 <= ey <= y1)

    def chord_lengths(self, c: float, axis: int) -> list[float]:
        """Lengths of the pieces of the line {coord[axis] == c} inside the closed polygon."""
        o = 1 - axis
        cuts: list[float] = []
        flats: list[tuple[float, float]] = []
        for ring in self.rings:
            a, b = ring[:-1], ring[1:]
            for k in range(len(a)):
                ca, cb = a[k][axis], b[k][axis]
                if ca == c and cb == c:                       # edge lies on the line
                    flats.append((min(a[k][o], b[k][o]), max(a[k][o], b[k][o])))
                elif (ca <= c) != (cb <= c):                  # half-open crossing rule
                    t = (c - ca) / (cb - ca)
                    cuts.append(a[k][o] + t * (b[k][o] - a[k][o]))
        cuts.sort()
        spans = [(cuts[i], cuts[i + 1]) for i in range(0, len(cuts) - 1, 2)] + flats
        if not spans:
            return []
        spans.sort()
        merged = [list(spans[0])]
        for s0, s1 in spans[1:]:
            if s0 <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], s1)
            else:
                merged.append([s0, s1])
        return [s1 - s0 for s0, s1 in merged]


def _ring_area(r: np.ndarray) -> float:
    return 0.5 * float(np.sum(r[:-1, 0] * r[1:, 1] - r[1:, 0] * r[:-1, 1]))


def _orient(a, b, c) -> float:
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


def _on_segment(a, b, p) -> bool:
    return min(a[0], b[0]) <= p[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= p[1] <= max(a[1], b[1])


def _segments_intersect(a, b, c, d) -> bool:
    """closed segments ab and cd share a point"""
    o1, o2, o3, o4 = _orient(a, b, c), _orient(a, b, d), _orient(c, d, a), _orient(c, d, b)
    if ((o1 > 0) != (o2 > 0)) and ((o3 > 0) != (o4 > 0)) and o1 != 0 and o2 != 0 and o3 != 0 and o4 != 0:
        return True
    return ((o1 == 0 and _on_segment(a, b, c)) or (o2 == 0 and _on_segment(a, b, d))
            or (o3 == 0 and _on_segment(c, d, a)) or (o4 == 0 and _on_segment(c, d, b)))


def _locate_in_ring(ring: np.ndarray, x: float, y: float) -> int:
    if len(ring) > 64:                                    # long rings (raster contours): numpy over the edges
        a, b = ring[:-1], ring[1:]
        orient = (b[:, 0] - a[:, 0]) * (y - a[:, 1]) - (b[:, 1] - a[:, 1]) * (x - a[:, 0])
        on = (orient == 0) & (np.minimum(a[:, 0], b[:, 0]) <= x) & (x <= np.maximum(a[:, 0], b[:, 0])) & \
             (np.minimum(a[:, 1], b[:, 1]) <= y) & (y <= np.maximum(a[:, 1], b[:, 1]))
        if on.any():
            return 0
        span = (a[:, 1] > y) != (b[:, 1] > y)
        with np.errstate(divide="ignore", invalid="ignore"):
            xi = a[:, 0] + (y - a[:, 1]) * (b[:, 0] - a[:, 0]) / (b[:, 1] - a[:, 1])
        return 1 if int((span & (xi > x)).sum()) % 2 == 1 else -1
    inside = False
    p = (x, y)
    for k in range(len(ring) - 1):
        a, b = ring[k], ring[k + 1]
        if _orient(a, b, p) == 0 and _on_segment(a, b, p):
            return 0
        if (a[1] > y) != (b[1] > y):
            xi = a[0] + (y - a[1]) * (b[0] - a[0]) / (b[1] - a[1])
            if xi > x:
                inside = not inside
    return 1 if inside else -1


def _ring_is_simple(ring: np.ndarray) -> bool:
    n = len(ring) - 1
    if n > 2000:          # O(n^2) check; huge hand-drawn ROIs are trusted
        return True
    for i in range(n):
        for j in range(i + 1, n):
            adjacent = j == i + 1 or (i == 0 and j == n - 1)
            if adjacent:
                continue
            if _segments_intersect(ring[i], ring[i + 1], ring[j], ring[j + 1]):
                return False
    return True


# ---------------------------------------------------------------------------------
def _edges(poly: "RoiPolygon"):
    a = np.concatenate([r[:-1] for r in poly.rings])
    b = np.concatenate([r[1:] for r in poly.rings])
    keep = a[:, 1] != b[:, 1]                       # horizontal edges carry no area in a y-sweep
    return a[keep], b[keep]


def _chords_length_overlap(ea, eb, y: float) -> float:
    """length of (A_y intersect B_y) at height y: even-odd x-intervals of both edge sets, merged"""
    def crossings(e):
        a, b = e
        m = (np.minimum(a[:, 1], b[:, 1]) < y) & (y < np.maximum(a[:, 1], b[:, 1]))
        t = (y - a[m, 1]) / (b[m, 1] - a[m, 1])
        return np.sort(a[m, 0] + t * (b[m, 0] - a[m, 0]))
    xa, xb = crossings(ea), crossings(eb)
    if len(xa) < 2 or len(xb) < 2:
        return 0.0
    ia, ib = xa[: len(xa) // 2 * 2].reshape(-1, 2), xb[: len(xb) // 2 * 2].reshape(-1, 2)
    lo = np.maximum(ia[:, None, 0], ib[None, :, 0])
    hi = np.minimum(ia[:, None, 1], ib[None, :, 1])
    return float(np.clip(hi - lo, 0.0, None).sum())


def intersection_area(pa: "RoiPolygon", pb: "RoiPolygon") -> float:
    """``pa.intersection(pb).area`` of shapely (predict_wsi.py:1822-1826: artefact area inside an ROI class)
    without constructing the intersection: sweep in y over the slabs between consecutive events (vertices of
    either polygon, crossings of an edge of one with an edge of the other).  Inside a slab no edge starts, ends
    or crosses, so the overlap length of the two even-odd chord sets is LINEAR in y and its integral is
    height x (length at mid-height) exactly.  Holes subtract through the even-odd rule."""
    ax0, ay0, ax1, ay1 = pa.bounds
    bx0, by0, bx1, by1 = pb.bounds
    ylo, yhi = max(ay0, by0), min(ay1, by1)
    if ylo >= yhi or max(ax0, bx0) >= min(ax1, bx1):
        return 0.0
    ea, eb = _edges(pa), _edges(pb)
    ys = [np.array([ylo, yhi])]
    for r in pa.rings + pb.rings:
        ys.append(r[:, 1])
    # edge(A) x edge(B) crossings, vectorised: solve a0 + t (a1 - a0) = b0 + u (b1 - b0)
    (a0, a1), (b0, b1) = ea, eb
    if len(a0) and len(b0):
        da, db = (a1 - a0)[:, None, :], (b1 - b0)[None, :, :]
        w = (b0[None, :, :] - a0[:, None, :])
        den = da[..., 0] * db[..., 1] - da[..., 1] * db[..., 0]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (w[..., 0] * db[..., 1] - w[..., 1] * db[..., 0]) / den
            u = (w[..., 0] * da[..., 1] - w[..., 1] * da[..., 0]) / den
        hit = (den != 0) & (t > 0) & (t < 1) & (u > 0) & (u < 1)
        if hit.any():
            ii = np.nonzero(hit)
            ys.append(a0[ii[0], 1] + t[hit] * (a1[ii[0], 1] - a0[ii[0], 1]))
    ys = np.unique(np.concatenate(ys))
    ys = ys[(ys >= ylo) & (ys <= yhi)]
    area = 0.0
    for y0, y1 in zip(ys[:-1], ys[1:]):
        if y1 > y0:
            area += (y1 - y0) * _chords_length_overlap(ea, eb, 0.5 * (y0 + y1))
    return area


def split_self_intersections(ring: np.ndarray, max_splits: int = 256) -> list[np.ndarray]:
    """Repair of a self-intersecting ring in the spirit of ``shapely.make_valid(polygon, method="structure")``
    (predict_wsi.py:1029-1047): the ring is noded at its proper self-crossings and cut into simple closed loops
    (a bow tie becomes its two triangles), each of which is then a valid polygon of the resulting MultiPolygon.
    Where loops overlap each other GEOS dissolves them into one area; here they stay separate parts (the point
    sets agree, a cell in the overlap can then match two parts -- the reference's own behaviour for overlapping
    ROI polygons, :1290-1305)."""
    r = np.asarray(ring, dtype=np.float64)[:, :2]
    if len(r) and np.array_equal(r[0], r[-1]):
        r = r[:-1]
    work, out = [r], []
    while work and max_splits > 0:
        cur = work.pop()
        n = len(cur)
        found = None
        for i in range(n):
            a, b = cur[i], cur[(i + 1) % n]
            for j in range(i + 2, n):
                if i == 0 and j == n - 1:
                    continue                                  # neighbours through the closing edge
                c, d = cur[j], cur[(j + 1) % n]
                o1, o2, o3, o4 = _orient(a, b, c), _orient(a, b, d), _orient(c, d, a), _orient(c, d, b)
                if o1 * o2 < 0 and o3 * o4 < 0:               # proper crossing
                    t = o3 / (o3 - o4)
                    found = (i, j, a + t * (b - a))
                    break
            if found:
                break
        if not found:
            if len(cur) >= 3 and abs(_ring_area(np.concatenate([cur, cur[:1]]))) > 0:
                out.append(np.concatenate([cur, cur[:1]]))
            continue
        i, j, pnt = found
        max_splits -= 1
        work.append(np.concatenate([cur[: i + 1], pnt[None], cur[j + 1:]]))      # outer remainder
        work.append(np.concatenate([pnt[None], cur[i + 1: j + 1]]))              # the loop that was cut off
    return out


def make_valid_polygon(exterior, holes=()) -> list["RoiPolygon"]:
    """``make_valid`` (predict_wsi.py:1050-1084) for one polygon: a valid polygon comes back as it is, an invalid
    exterior is cut into its simple loops (see ``split_self_intersections``); holes go to the loop that contains
    their first vertex, invalid holes are dropped."""
    try:
        return [RoiPolygon(exterior, holes)]
    except ValueError:
        pass
    parts = []
    good_holes = []
    for h in holes:
        try:
            good_holes.append(RoiPolygon(h).exterior)
        except ValueError:
            continue
    for loop in split_self_intersections(exterior):
        shell = RoiPolygon(loop, validate=False)
        mine = [h for h in good_holes if shell.contains_point_strict(float(h[0, 0]), float(h[0, 1]))]
        parts.append(RoiPolygon(loop, mine, validate=False))
    if not parts:
        raise ValueError("ROI ring is degenerate (no area)")
    return parts


def load_roi_polygons(roi_geojson_path: str, group_by_class: bool = False):
    """GeoJSON FeatureCollection / feature list / single feature -> list[RoiPolygon] (or None when
    empty); with ``group_by_class`` also ``{classification.name: [polygons]}`` like the reference.
    MultiPolygons are flattened, a LineString is closed into a polygon (:1128-1145)."""
    with open(roi_geojson_path, "r") as f:
        data = json.load(f)
    if isinstance(data, list):
        data = {"features": data}
    if "features" not in data and "geometry" in data:
        data["features"] = [data]
    polys: list[RoiPolygon] = []
    class_dict: dict[str, list[RoiPolygon]] = {}
    for feat in data.get("features", []):
        geom = feat.get("geometry")
        if not geom:
            continue
        kind, coords = geom.get("type"), geom.get("coordinates")
        if kind == "Polygon":
            parts = make_valid_polygon(coords[0], coords[1:])
        elif kind == "MultiPolygon":
            parts = [p for c in coords for p in make_valid_polygon(c[0], c[1:])]
        elif kind == "LineString":
            parts = make_valid_polygon(coords)
        else:
            continue                                           # points etc. are ignored by the reference too
        name = feat.get("properties", {}).get("classification", {}).get("name", "unknown")
        for p in parts:
            polys.append(p)
            if group_by_class:
                class_dict.setdefault(name, []).append(p)
    if not polys:
        return None
    return (polys, class_dict) if group_by_class else polys


def get_maximum_lengths(polygon: RoiPolygon, n_samples: int = 100) -> tuple[float, float]:
    """Maximum chord of ``n_samples`` vertical and of ``n_samples`` horizontal scan lines."""
    x, y, x_max, y_max = polygon.bounds
    lx = [l for c in np.linspace(x, x_max, n_samples) for l in polygon.chord_lengths(float(c), 0) if l > 0]
    ly = [l for c in np.linspace(y, y_max, n_samples) for l in polygon.chord_lengths(float(c), 1)]
    return max(lx), max(ly)


def get_coords_roi(rois: list[RoiPolygon], tile_size: int, overlap: int, model_overlap: int, ts: float):
    """``_get_coords_roi``: per ROI a grid over its (padded) bounding box with the tile side
    clamped to [256, tile_size] by the ROI's narrowest maximum chord; the last row/column is
    pulled back inside.  ``tile_size``/``overlap`` are the read (level) sizes, ``model_overlap``
    is ``self.overlap`` (the reference pads the box by ``self.overlap // 2``)."""
    out = []
    adj = model_overlap // 2
    for geom in rois:
        coords = geom.exterior.astype(int)
        coords = (coords / ts).astype(int)
        cmin, cmax = coords.min(axis=0) - adj, coords.max(axis=0) + adj
        min_max_lens = int(min(get_maximum_lengths(geom)) / ts)
        cts = min(max(min_max_lens, MIN_TILE_SIZE), tile_size)
        for i in range(int(cmin[0]), int(cmax[0]), cts - overlap):
            if (i + cts) > cmax[0]:
                i = int(cmax[0]) - cts
            for j in range(int(cmin[1]), int(cmax[1]), cts - overlap):
                if (j + cts) > cmax[1]:
                    j = int(cmax[1]) - cts
                out.append(((int(i * ts), int(j * ts)), cts))
    return out


def check_tile_in_cnts(coords, tile_size: int, ts: float, cnts: list[RoiPolygon]) -> bool:
    size0 = tile_size * float(ts)
    return any(c.intersects_square(coords[0], coords[1], size0) for c in cnts)


def polygon_to_geojson(polygon: RoiPolygon, id: str | None = None, object_type: str = "annotation",
                       additional_properties: dict | None = None) -> list[dict]:
    """``shapely_polygon_to_geojson`` (predict_wsi.py:966-1023) for one polygon with holes"""
    import uuid
    cx, cy = polygon.centroid
    props = {"objectType": object_type, "isLocked": False,
             "measurements": [{"name": "area", "value": polygon.area}, {"name": "perimeter", "value": polygon.length},
                              {"name": "centroidX", "value": cx}, {"name": "centroidY", "value": cy}]}
    if additional_properties is not None:
        props.update(additional_properties)
    return [{"type": "Feature", "id": id if id is not None else str(uuid.uuid4()),
             "geometry": {"type": "Polygon", "coordinates": [r.tolist() for r in polygon.rings]},
             "properties": props}]


def polygons_from_cnts(cnts: dict) -> list[RoiPolygon]:
    """{"contour", "holes"} dicts of detect_tissue_wsi / detect_artefacts_wsi -> polygons
    (``make_valid(shapely.Polygon(cnt["contour"], cnt["holes"]))``, predict_wsi.py:314-319, and
    ``create_valid_polygon``, outputs.py:17-55); degenerate contours are dropped."""
    out = []
    for c in cnts.values():
        try:
            poly = RoiPolygon(c["contour"], c.get("holes", []), validate=False)
        except ValueError:
            continue
        if poly.area > 0:
            out.append(poly)
    return out


def filter_cells_by_artefacts(cells: list[dict], artefact_polys: list[RoiPolygon]):
    """predict_wsi.py:1281-1333: drop cells whose centroid lies strictly inside an artefact polygon"""
    if not artefact_polys:
        return cells, 0
    if len(cells) == 0:
        return cells, 0
    pts = np.array([get_cell_centroid(cell) for cell in cells], dtype=np.float64)
    hit = np.zeros(len(cells), bool)
    for c in artefact_polys:
        hit |= c.contains_points_strict(pts[:, 0], pts[:, 1])
    return [cell for cell, h in zip(cells, hit) if not h], int(hit.sum())


def get_artefact_class_id(class_name: str) -> int:
    """predict_wsi.py:1203-1221"""
    return {"Fold": 2, "Darkspot & Foreign Object": 3, "PenMarking": 4, "Edge & Air Bubble": 5, "OOF": 6}.get(class_name, 0)


def get_cell_centroid(cell: dict) -> list[float]:
    m = cell["properties"]["measurements"]
    return [[x for x in m if x["name"] == "centroidX"][0]["value"],
            [x for x in m if x["name"] == "centroidY"][0]["value"]]


def filter_cells_by_contours(polygons: list[dict], contours: list[RoiPolygon]) -> list[dict]:
    """Keep cells whose centroid lies strictly inside a contour.  Like the reference's
    ``STRtree.query(points, predicate="within")`` a cell inside k overlapping contours is
    returned k times."""
    if len(contours) == 0:
        return polygons
    if len(polygons) == 0:
        return polygons
    pts = np.array([get_cell_centroid(cell) for cell in polygons], dtype=np.float64)
    counts = np.zeros(len(polygons), np.int64)
    for c in contours:
        counts += c.contains_points_strict(pts[:, 0], pts[:, 1])
    return [polygons[i] for i in range(len(polygons)) for _ in range(int(counts[i]))]
