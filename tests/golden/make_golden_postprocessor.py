"""Mint a golden vector from the reference's OWN per-tile polygon loop (build container only):
``PostProcessor.__call__``  /root/reference/src/classpose/entrypoints/predict_wsi.py:578-656, run unbound on a fake
``self`` (its constructor starts a process and queues), on the id / class maps that the reference's ``eval`` produced
(tests/golden/reference_eval.npz).  What lives in the ABSENT wheels is supplied at the call boundary by the oracle's
restatements -- ``cv2.findContours(..., RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)`` = oracle/polygons.find_contours_external_simple
in OpenCV's (n, 1, 2) shape, ``shapely.Polygon`` = a shim over the oracle's GEOS-order ring metrics -- so those stay
"unpinned"; what the vector PINS is the loop the reference owns: ``find_objects`` slices and label order, ``contours[0]``,
bbox offset then ``* scale + coords``, the < 4 vertices and validity rejections and their counter, the centroid rounded to
2 decimals, the closed ring, the class taken from the cell's first raster pixel and ``labels[cl - 1]`` / ``COLORMAP[cl - 1]``
(Python's negative index for class 0), and the keys of the cell dict.  Fixtures hold data only.
        python tests/golden/make_golden_postprocessor.py
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402

LABELS = ["Alpha", "Beta", "Gamma"]
CASES = ((0, (1000, 2000), 1.0, True), (1, (33, 7), 2.015625, True), (0, (0, 0), 0.5, False))   # (eval case, coords, scale, labelled)


def main():
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, mg.REF)
    from oracle import polygons as op
    from classpose_amd import geojson as our_geojson
    import classpose.entrypoints.predict_wsi as pw

    class Poly:                                               # shapely.Polygon over the oracle's ring metrics
        def __init__(self, coords):
            m = op.polygon_metrics(np.asarray(coords, dtype=np.float64))
            self.is_valid, self.area, self.length = m["valid"], m["area"], m["perimeter"]
            self.centroid = types.SimpleNamespace(coords=[tuple(m["centroid"])])

    def find_contours(img, mode, method):
        assert (mode, method) == ("EXT", "SIMPLE") and img.dtype == np.uint8
        return ([c.reshape(-1, 1, 2) for c in op.find_contours_external_simple(img)], None)

    pw.cv2.findContours, pw.cv2.RETR_EXTERNAL, pw.cv2.CHAIN_APPROX_SIMPLE = find_contours, "EXT", "SIMPLE"
    pw.shapely.Polygon = Poly
    pw.COLORMAP = our_geojson.COLORMAP                        # matplotlib Set3 x 255 (a constant of an absent wheel)

    ev = np.load(os.path.join(HERE, "reference_eval.npz"))
    out = {}
    for k, (case, coords, scale, labelled) in enumerate(CASES):
        masks = ev[f"ev_{case}_masks"].astype(np.uint16)
        cm = ev[f"ev_{case}_class_masks"].astype(np.int64)
        if k == 1:                                            # exercise the rejections too: a 1-pixel and a 2-pixel instance
            masks = masks.copy()
            n = int(masks.max())
            masks[0, 0] = n + 1
            masks[5, 250:252] = n + 2
            cm = cm.copy(); cm[0, 0] = 1; cm[5, 250:252] = 2
        sink = []
        me = types.SimpleNamespace(labels=LABELS if labelled else None, n_invalid_cells=types.SimpleNamespace(value=0),
                                   n_cells=types.SimpleNamespace(value=0), value=types.SimpleNamespace(value=0),
                                   polygons=types.SimpleNamespace(put=sink.append))
        datum = (masks, cm) if labelled else masks
        pw.PostProcessor.__call__(me, [datum], [np.array(coords)], scale)
        cells = sink[0]
        assert me.n_cells.value == len(cells) and me.value.value == 1
        out[f"pp_{k}_cfg"] = np.array([case, coords[0], coords[1], int(labelled)])
        out[f"pp_{k}_scale"] = np.array(scale)
        out[f"pp_{k}_extra"] = np.array(int(k == 1))
        out[f"pp_{k}_n_invalid"] = np.array(me.n_invalid_cells.value)
        out[f"pp_{k}_keys"] = np.array(json.dumps(sorted(cells[0].keys())))
        out[f"pp_{k}_n_pts"] = np.array([len(c["coords"]) for c in cells])
        out[f"pp_{k}_coords"] = np.concatenate([np.asarray(c["coords"], dtype=np.float64) for c in cells])
        out[f"pp_{k}_area"] = np.array([c["area"] for c in cells])
        out[f"pp_{k}_perimeter"] = np.array([c["perimeter"] for c in cells])
        out[f"pp_{k}_centroid"] = np.array([c["centroid"] for c in cells])
        out[f"pp_{k}_class_int"] = np.array([c["class_int"] for c in cells])
        out[f"pp_{k}_label"] = np.array(json.dumps([c["label"] for c in cells]))
        out[f"pp_{k}_color"] = np.array([c["color"] for c in cells])
        print(k, len(cells), "cells,", me.n_invalid_cells.value, "invalid, keys", sorted(cells[0].keys()))
    out["pp_n"] = np.array(len(CASES))
    np.savez_compressed(os.path.join(HERE, "reference_postprocessor.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "reference_postprocessor.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
