"""Tabular outputs of the WSI path: per-class cell counts and densities (``--output_type csv``).

Mirrors ``calculate_cellular_densities`` and ``map_cells_to_roi_classes``
(/root/reference/src/classpose/entrypoints/outputs.py:57-273) and the way ``main`` feeds them
(predict_wsi.py:1786-1857).  ``--output_type spatialdata`` needs the spatialdata / geopandas
stack (absent from this image) and is not built.
"""
from __future__ import annotations

import pandas as pd

from . import roi
from .log import get_logger

logger = get_logger("classpose")


def map_cells_to_roi_classes(cells: list[dict], roi_class_dict: dict[str, list], priority_list=None):
    """cell -> the first ROI class (priority order, then dict order) with a polygon that strictly
    contains the centroid of the cell's polygon"""
    if priority_list:
        invalid = [c for c in priority_list if c not in roi_class_dict]
        if invalid:
            logger.warning(f"Priority list contains classes not found in ROI: {invalid}")
        ordered = [c for c in priority_list if c in roi_class_dict] + \
                  [c for c in roi_class_dict if c not in priority_list]
    else:
        ordered = list(roi_class_dict.keys())
    result = {name: [] for name in roi_class_dict}
    for cell in cells:
        try:
            poly = roi.RoiPolygon(cell["geometry"]["coordinates"][0], validate=False)
        except ValueError:
            continue
        cx, cy = poly.centroid
        for name in ordered:
            if any(p.contains_point_strict(cx, cy) for p in roi_class_dict.get(name, [])):
                result[name].append(cell)
                break
    for name, cl in result.items():
        logger.info(f"ROI class '{name}': {len(cl)} cells")
    return result


def map_points_to_roi_classes(xs, ys, roi_class_dict: dict[str, list], priority_list=None):
    """array form of ``map_cells_to_roi_classes``: region name per point (None when no class contains it)"""
    import numpy as np
    if priority_list:
        ordered = [c for c in priority_list if c in roi_class_dict] + \
                  [c for c in roi_class_dict if c not in priority_list]
    else:
        ordered = list(roi_class_dict.keys())
    region = np.full(len(xs), None, dtype=object)
    for name in ordered:
        hit = np.zeros(len(xs), bool)
        for p in roi_class_dict.get(name, []):
            hit |= p.contains_points_strict(xs, ys)
        region[hit & (region == None)] = name            # noqa: E711  (object array comparison)
    return region


def densities_from_counts(region: str, counts: dict, tissue_area_pixels, artefact_area_pixels, mpp_x: float,
                          mpp_y: float, labels: list[str]) -> pd.DataFrame:
    eff_mm2 = (tissue_area_pixels - artefact_area_pixels) * (mpp_x * mpp_y) / 1e6
    return pd.DataFrame([{"region": region, "cell_class": l, "count": counts.get(l, 0),
                          "density": counts.get(l, 0) / eff_mm2 if eff_mm2 > 0 else 0} for l in labels])


def calculate_cellular_densities(cells, tissue_area_pixels, artefact_area_pixels, mpp_x: float, mpp_y: float,
                                 labels: list[str]) -> pd.DataFrame:
    """rows (region, cell_class, count, density [cells/mm^2]); density = count / ((tissue - artefact)
    area * mpp_x * mpp_y / 1e6), 0 when the effective area is not positive"""
    mpp_product = mpp_x * mpp_y
    rows = []

    def block(region, region_cells, tissue, artefact):
        eff_mm2 = (tissue - artefact) * mpp_product / 1e6
        counts = {label: 0 for label in labels}
        for cell in region_cells:
            name = cell["properties"]["classification"]["name"]
            if name in counts:
                counts[name] += 1
        for label in labels:
            rows.append({"region": region, "cell_class": label, "count": counts[label],
                         "density": counts[label] / eff_mm2 if eff_mm2 > 0 else 0})

    if isinstance(cells, dict):
        for region in cells.keys():
            block(region, cells[region], tissue_area_pixels.get(region, 0), artefact_area_pixels.get(region, 0))
    else:
        block("tissue", cells, tissue_area_pixels, artefact_area_pixels)
    return pd.DataFrame(rows)
