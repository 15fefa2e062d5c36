"""Cells -> QuPath GeoJSON (host side), cross-tile de-duplication.

Mirrors /root/reference/src/classpose/entrypoints/predict_wsi.py:
``to_geojson_polygon`` :813-854, ``apply_bounds_offset_to_feature`` :857-893,
``deduplicate`` :896-965 (KDTree pairs within 7.5 px, greedy order-dependent grouping over
the ``set`` of pairs, keep the largest area), ``polygons_to_centroids`` :1336-1374 and the
colour table ``COLORMAP`` :99 (matplotlib ``Set3`` x 255, truncated to int).
"""
from __future__ import annotations

import uuid

import numpy as np
from scipy.spatial import KDTree

COLORMAP = [[141, 211, 199], [255, 255, 179], [190, 186, 218], [251, 128, 114], [128, 177, 211],
            [253, 180, 98], [179, 222, 105], [252, 205, 229], [217, 217, 217], [188, 128, 189],
            [204, 235, 197], [255, 237, 111]]


def cell_dict(coords: list, cl: int, labels: list[str] | None, area: float, perimeter: float,
              centroid: list) -> dict:
    """One entry of ``curr_cells`` (predict_wsi.py:629-652). ``cl`` is the class-map value;
    class 0 indexes labels[-1] / COLORMAP[-1] exactly like the reference's ``labels[cl - 1]``."""
    if labels is not None:
        label, color, class_int = labels[int(cl) - 1], COLORMAP[int(cl) - 1], int(cl) - 1
    else:
        label, color, class_int = "cell", [0, 168, 132], 0
    coords = [list(map(float, p)) for p in coords]
    coords.append(coords[0].copy())
    return {"id": str(uuid.uuid4()), "coords": coords, "class_int": class_int, "area": float(area),
            "label": label, "color": color, "perimeter": float(perimeter), "centroid": centroid}


def to_geojson_polygon(curr_cell: dict) -> dict:
    return {
        "type": "Feature",
        "id": curr_cell["id"],
        "geometry": {"type": "Polygon", "coordinates": [curr_cell["coords"]]},
        "properties": {
            "objectType": "annotation",
            "isLocked": False,
            "classification": {"name": curr_cell["label"], "color": curr_cell["color"]},
            "measurements": [
                {"name": "area", "value": curr_cell["area"]},
                {"name": "perimeter", "value": curr_cell["perimeter"]},
                {"name": "centroidX", "value": curr_cell["centroid"][0]},
                {"name": "centroidY", "value": curr_cell["centroid"][1]},
            ],
        },
    }


def apply_bounds_offset_to_feature(feature: dict, bounds_x: float, bounds_y: float) -> dict:
    if not feature or "geometry" not in feature or "coordinates" not in feature["geometry"]:
        return feature
    g = feature["geometry"]
    g["coordinates"] = [[[p[0] - bounds_x, p[1] - bounds_y] for p in ring] for ring in g["coordinates"]]
    for m in feature.get("properties", {}).get("measurements", []):
        if m["name"] == "centroidX":
            m["value"] -= bounds_x
        elif m["name"] == "centroidY":
            m["value"] -= bounds_y
    return feature


def _measure(feature: dict, name: str):
    return [x for x in feature["properties"]["measurements"] if x["name"] == name][0]["value"]


def dedup_indices(centers, sizes, max_dist: float = 15 / 2) -> list[int]:
    """Indices kept by the reference's ``deduplicate`` (predict_wsi.py:896-965): every pair of scipy's set, in the set's own
    order, through the grouping loop -- the plain form ``dedup_exact`` is checked against."""
    if len(centers) == 0:
        return []
    keep = np.ones(len(centers), bool)
    _group_pairs_in_order(KDTree(centers).query_pairs(max_dist), np.asarray(sizes), keep)
    return np.flatnonzero(keep).tolist()


_XXP1, _XXP2, _XXP5 = np.uint64(11400714785074694791), np.uint64(14029467366897019727), np.uint64(2870177450012600261)


def _tuple2_hash(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """CPython >= 3.8 ``hash((a, b))`` for non-negative ints below 2**61 - 1 (tupleobject.c, xxHash-style), vectorised."""
    with np.errstate(over="ignore"):
        acc = np.full(a.shape, _XXP5, np.uint64)
        for lane in (a.astype(np.uint64), b.astype(np.uint64)):
            acc = acc + lane * _XXP2
            acc = (acc << np.uint64(31)) | (acc >> np.uint64(33))
            acc = acc * _XXP1
        acc = acc + (np.uint64(2) ^ (_XXP5 ^ np.uint64(3527539)))
    return acc


def _set_table_mask(n: int) -> int:
    """mask of a CPython ``set`` after n distinct ``add`` calls (setobject.c: grow when fill*5 >= mask*3, to the
    first power of two above 4x (2x beyond 50 000) the number of entries)"""
    mask = 7
    for used in range(1, n + 1):
        if used * 5 >= mask * 3:
            minused = used * 2 if used > 50000 else used * 4
            size = 8
            while size <= minused:
                size <<= 1
            mask = size - 1
    return mask


def _fast_set_table_mask(n: int) -> int:
    mask, used = 7, 0
    while True:
        # entries at which the next resize happens with the current mask
        trig = -(-(mask * 3) // 5)
        if trig > n:
            return mask
        used = trig
        minused = used * 2 if used > 50000 else used * 4
        size = 8
        while size <= minused:
            size <<= 1
        mask = size - 1


def _group_pairs_in_order(ordered, sizes, keep) -> None:
    """The reference's grouping loop (predict_wsi.py:929-960) over pairs visited in the given order; clears ``keep``
    for every cell it removes."""
    groups: dict[int, list] = {}
    member_to_group: dict[int, int] = {}
    for pair in ordered:
        if pair[0] not in member_to_group and pair[1] not in member_to_group:
            gi = len(groups)
            groups[gi] = []
            member_to_group[pair[0]] = gi
            member_to_group[pair[1]] = gi
        else:
            gi = member_to_group[pair[0]] if pair[0] in member_to_group else member_to_group[pair[1]]
        if pair[0] not in groups[gi]:
            groups[gi].append(pair[0])
        if pair[1] not in groups[gi]:
            groups[gi].append(pair[1])
    szl = sizes.tolist() if hasattr(sizes, "tolist") else list(sizes)
    for group in groups.values():
        if len(group) > 1:
            largest = group[int(np.argmax([szl[i] for i in group]))]
            for i in group:
                if i != largest:
                    keep[i] = False


def _split_simple(n: int, sizes, pi, pj, keep):
    """Resolves the connected components of exactly two cells (the bulk: one cell seen by two overlapping tiles) --
    they do not depend on any visiting order: group [i, j] with i < j, ``np.argmax`` keeps i on equal areas -- and
    returns the mask of the pairs that belong to larger components."""
    deg = np.bincount(pi, minlength=n) + np.bincount(pj, minlength=n)
    simple = (deg[pi] == 1) & (deg[pj] == 1)
    si, sj = pi[simple], pj[simple]
    drop_j = sizes[si] >= sizes[sj]                      # np.argmax([s_i, s_j]) == 0 unless s_j is strictly larger
    keep[np.where(drop_j, sj, si)] = False
    return ~simple


def count_order_dependent(n: int, pairs: np.ndarray) -> int:
    """Number of cells that sit in connected components of three or more cells of the neighbour graph: the cells whose
    fate in the reference's ``deduplicate`` depends on the iteration order of its Python set of pairs."""
    if len(pairs) == 0:
        return 0
    pi, pj = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    deg = np.bincount(pi, minlength=n) + np.bincount(pj, minlength=n)
    cx = ~((deg[pi] == 1) & (deg[pj] == 1))
    return int(len(np.unique(np.concatenate([pi[cx], pj[cx]]))))


def dedup_exact(centers, sizes, max_dist: float = 15 / 2, stats: dict | None = None) -> np.ndarray:
    """Indices kept by the reference's ``deduplicate`` (predict_wsi.py:896-965), result-identical to it, at array speed.

    The neighbour pairs come from the same call as in the reference -- ``KDTree(centers).query_pairs(max_dist)``, a
    Python ``set`` of (i, j) tuples with i < j -- and are visited in THAT set's iteration order, which is what decides
    the outcome inside connected components of three or more cells (4x corner overlaps, chains).  Components never
    interact in the reference's loop (a pair only ever joins the group of one of its own two cells), and components
    of exactly two cells do not depend on the order at all, so those are resolved vectorised and only the pairs of the
    larger components run through the loop, in their relative set order.  ``tests/test_dedup.py`` checks equality
    with the verbatim loop (``dedup_indices``)."""
    n = len(centers)
    keep = np.ones(n, bool)
    if n == 0:
        return np.flatnonzero(keep)
    sizes = np.asarray(sizes)
    neighbours = KDTree(np.asarray(centers, dtype=np.float64)).query_pairs(max_dist)
    if stats is not None:
        stats["n_pairs"] = len(neighbours)
        stats["n_order_dependent"] = 0
    if not neighbours:
        return np.flatnonzero(keep)
    from itertools import chain
    arr = np.fromiter(chain.from_iterable(neighbours), np.int64, 2 * len(neighbours)).reshape(-1, 2)   # set order kept
    pi, pj = arr[:, 0], arr[:, 1]
    cx = _split_simple(n, sizes, pi, pj, keep)
    if cx.any():
        if stats is not None:
            stats["n_order_dependent"] = int(len(np.unique(np.concatenate([pi[cx], pj[cx]]))))
        _group_pairs_in_order(zip(pi[cx].tolist(), pj[cx].tolist()), sizes, keep)
    return np.flatnonzero(keep)


def dedup_from_pairs(n: int, sizes, pairs: np.ndarray) -> np.ndarray:
    """FAST, APPROXIMATE-IN-CLUSTERS variant of ``deduplicate`` (predict_wsi.py:929-965) given the neighbour pairs
    (i < j) from the device radius search (``ops.dedup_pairs``; opt-in: ``CLASSPOSE_DEDUP_BACKEND=device``).

    The pair SET is exact.  The reference walks its Python ``set`` of pairs in set order, so inside connected components
    of three or more cells its result depends on that order.  Components of exactly two cells are order-free and are
    resolved vectorised; the pairs of the larger components are visited in the order of their slot in the hash table
    the reference's set would have (``hash((i, j)) & mask`` with the mask of a set holding ALL pairs): that IS the
    set's iteration order except for entries displaced by hash collisions, whose position also depends on scipy's
    insertion order (which no pair list carries) -- a fraction of a percent of the cells of such clusters can differ
    from the reference.  ``dedup_exact`` is the default and has no such caveat."""
    sizes = np.asarray(sizes)
    keep = np.ones(n, bool)
    if len(pairs) == 0:
        return np.flatnonzero(keep)
    pi, pj = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    cx = _split_simple(n, sizes, pi, pj, keep)
    if cx.any():
        ci, cj = pi[cx], pj[cx]
        slot = _tuple2_hash(ci, cj) & np.uint64(_fast_set_table_mask(len(pairs)))
        order = np.argsort(slot, kind="stable")
        _group_pairs_in_order(zip(ci[order].tolist(), cj[order].tolist()), sizes, keep)
    return np.flatnonzero(keep)


def deduplicate(features: list[dict], max_dist: float = 15 / 2) -> list[dict]:
    centers = [[_measure(f, "centroidX"), _measure(f, "centroidY")] for f in features]
    sizes = [_measure(f, "area") for f in features]
    return [features[i] for i in dedup_indices(centers, sizes, max_dist)]


def polygons_to_centroids(cells: list[dict]) -> list[dict]:
    out = []
    for cell in cells:
        out.append({
            "type": "Feature",
            "id": str(uuid.uuid4()),
            "geometry": {"type": "Point",
                         "coordinates": [_measure(cell, "centroidX"), _measure(cell, "centroidY")]},
            "properties": {"objectType": "annotation", "isLocked": False,
                           "classification": cell["properties"]["classification"],
                           "measurements": cell["properties"]["measurements"]},
        })
    return out


# ---------------------------------------------------------------------------------------------
# struct-of-arrays path of the CLI: the same features as the dict functions above, serialised
# straight from the cell table (no per-cell Python dicts are held: a 40k^2 slide has ~10^6 cells)
# ---------------------------------------------------------------------------------------------
def gather_vertices(xy: np.ndarray, offsets: np.ndarray, n_pts: np.ndarray) -> np.ndarray:
    """rows xy[offsets[i] : offsets[i] + n_pts[i]] of every i, concatenated (vectorised)"""
    n_pts = np.asarray(n_pts, dtype=np.int64)
    total = int(n_pts.sum())
    if total == 0:
        return np.zeros((0, 2), xy.dtype)
    starts = np.repeat(np.asarray(offsets, dtype=np.int64) - np.concatenate([[0], np.cumsum(n_pts)[:-1]]), n_pts)
    return xy[starts + np.arange(total)]


def rounded_centroids(cells: np.ndarray) -> np.ndarray:
    """``np.round(polygon.centroid.coords[0], 2)`` (predict_wsi.py:629) for the whole table"""
    return np.stack([np.round(cells["cx"], 2), np.round(cells["cy"], 2)], 1)


def class_json_table(labels: list[str] | None, n_values: int) -> list[str]:
    """The serialised ``classification`` object per class-map value ``cl`` (``labels[cl - 1]`` / ``COLORMAP[cl - 1]``
    with Python's negative index for ``cl == 0``, predict_wsi.py:629-652), as ``json.dumps`` writes it."""
    import json
    out = []
    for cl in range(max(int(n_values), 1)):
        if labels is not None:
            name, color = labels[cl - 1], COLORMAP[cl - 1]
        else:
            name, color = "cell", [0, 168, 132]
        out.append('{"name": %s, "color": %s}' % (json.dumps(name), json.dumps(color)))
    return out


def write_feature_collections(contours_path, centroids_path, cells: np.ndarray, xy: np.ndarray, keep,
                              labels: list[str] | None, bounds=(0.0, 0.0), n_threads: int = 0) -> int:
    """Writes ``{"type": "FeatureCollection", "features": [...]}`` for the cells ``keep`` (indices into
    the table, repeats allowed) exactly as ``json.dump`` of ``to_geojson_polygon`` /
    ``polygons_to_centroids`` features (after ``apply_bounds_offset_to_feature``) would.  The text is produced
    by the native streaming writer ``cpx_write_geojson`` (csrc/cpx_host_geojson.cpp): a Python loop over 2 M cells
    took 28 s at 40k x 40k, longer than the tile loop on 8 GPUs."""
    import ctypes as C

    from . import _lib
    L = _lib.lib()
    cells = np.ascontiguousarray(cells)
    assert cells.dtype.itemsize == 48, "cell table rows are {f8 area, perimeter, cx, cy; i8 n_pts, cls}"
    n = len(cells)
    keep = np.ascontiguousarray(np.asarray(keep, dtype=np.int64).reshape(-1))
    offs = np.ascontiguousarray(np.concatenate([[0], np.cumsum(cells["n_pts"])]).astype(np.int64))
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    cen = np.ascontiguousarray(rounded_centroids(cells), dtype=np.float64) if n else np.zeros((0, 2))
    n_cls = int(cells["cls"].max()) + 1 if n else 1
    if labels is not None and n and int(cells["cls"].min()) < 0:
        raise ValueError("negative class value in the cell table")
    table = [t.encode("ascii") for t in class_json_table(labels, n_cls)]
    arr = (C.c_char_p * len(table))(*table)
    _lib.check(L.cpx_write_geojson(str(contours_path).encode(), str(centroids_path).encode(),
                                   cells.ctypes.data, n, cen.ctypes.data, xy.ctypes.data, offs.ctypes.data,
                                   keep.ctypes.data, len(keep), arr, len(table), float(bounds[0]), float(bounds[1]),
                                   int(n_threads)), "write_geojson")
    return len(keep)
