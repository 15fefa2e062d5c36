"""ROI handling of the WSI path (host side): GeoJSON ROI loading, ROI-driven tile grid,
tile/ROI intersection test and the final centroid-in-ROI cell filter.

Mirrors ``load_roi_polygons`` (/root/reference/src/classpose/entrypoints/predict_wsi.py:1086-1161),
``get_maximum_lengths`` (:1164-1200), ``SlideLoader._get_coords_roi`` (:324-364),
``_check_tile_in_cnts`` (:406-417), ``_align_roi_tree_to_slide_bounds`` (:280-304) and
``filter_cells_by_contours`` (:1247-1278).  The reference delegates the geometry to
shapely/GEOS (absent from this image); the predicates it uses -- ``intersects`` between a
polygon and an axis-aligned square, point ``within`` polygon, chord lengths of axis-parallel
lines -- are restated here on float64 with the same open/closed conventions (``intersects`` is
closed, ``within`` excludes the boundary).  ``make_valid`` is NOT restated: an ROI ring that
self-intersects raises instead of being silently repaired differently from GEOS.
"""
from __future__ import annotations

import json

import numpy as np

MIN_TILE_SIZE = 256


class RoiPolygon:
    """One polygon: closed exterior ring + closed hole rings, float64 (n, 2) arrays."""

    def __init__(self, exterior, holes=()):
        self.exterior = self._close(np.asarray(exterior, dtype=np.float64)[:, :2])
        self.holes = [self._close(np.asarray(h, dtype=np.float64)[:, :2]) for h in holes]
        for ring in self.rings:
            if len(ring) < 4:
                raise ValueError("ROI ring needs at least three distinct vertices")
            if not _ring_is_simple(ring):
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

    def translate(self, xoff: float, yoff: float) -> "RoiPolygon":
        off = np.array([xoff, yoff])
        return RoiPolygon(self.exterior + off, [h + off for h in self.holes])

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

    def intersects_square(self, x0: float, y0: float, size: float) -> bool:
        """shapely ``polygon.intersects(square)`` (closed sets share at least one point)."""
        x1, y1 = x0 + size, y0 + size
        bx0, by0, bx1, by1 = self.bounds
        if bx1 < x0 or bx0 > x1 or by1 < y0 or by0 > y1:
            return False
        sq = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1], [x0, y0]])
        for ring in self.rings:
            for k in range(len(ring) - 1):
                ax, ay, bx, by = *ring[k], *ring[k + 1]
                if max(ax, bx) < x0 or min(ax, bx) > x1 or max(ay, by) < y0 or min(ay, by) > y1:
                    continue
                for j in range(4):
                    if _segments_intersect(ring[k], ring[k + 1], sq[j], sq[j + 1]):
                        return True
        # no boundary crossing: one contains the other, or they are disjoint
        if self.locate(x0, y0) >= 0:
            return True
        ex, ey = self.exterior[0]
        return x0 <= ex <= x1 and y0 <= ey <= y1

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
            parts = [RoiPolygon(coords[0], coords[1:])]
        elif kind == "MultiPolygon":
            parts = [RoiPolygon(c[0], c[1:]) for c in coords]
        elif kind == "LineString":
            parts = [RoiPolygon(coords)]
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
    boxes = np.array([c.bounds for c in contours])
    keep = []
    for i, cell in enumerate(polygons):
        x, y = get_cell_centroid(cell)
        cand = np.nonzero((boxes[:, 0] <= x) & (x <= boxes[:, 2]) & (boxes[:, 1] <= y) & (y <= boxes[:, 3]))[0]
        for k in cand:
            if contours[k].contains_point_strict(x, y):
                keep.append(i)
    return [polygons[i] for i in keep]
