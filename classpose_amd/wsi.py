"""Slide access and tile grid for the WSI path (host side).

Mirrors ``SlideLoader._init_slide`` / ``_get_coords`` / ``get_slide_resolution``
(/root/reference/src/classpose/entrypoints/predict_wsi.py:220-278,366-391 and
utils.py:285-331) and the ``WSIReader`` plugin switch
(/root/reference/src/classpose/__init__.py:6-41): readers are duck-typed on the
OpenSlide protocol; ``synthetic://`` URIs open the procedural slide of
``classpose_amd.synth`` (there are no slide files or OpenSlide wheels in the
build environment).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

WSI_READERS = ["czi-zeiss", "openslide", "synthetic"]


def get_wsi_reader(reader_str: str):
    if reader_str not in WSI_READERS:
        raise ValueError(f"Reader {reader_str} not supported. Should be one of {WSI_READERS}")
    if reader_str == "synthetic":
        from .synth import SyntheticSlide
        return SyntheticSlide.from_uri
    if reader_str == "czi-zeiss":
        raise ImportError("czi-zeiss reader requires pylibCZIrw (vendor library, not bundled)")
    from openslide import OpenSlide          # noqa: F401  (only when the wheel is installed)
    return OpenSlide


def WSIReader(path: str):
    """Env ``WSI_READER`` picks the reader class like the reference; synthetic:// overrides."""
    if str(path).startswith("synthetic://"):
        return get_wsi_reader("synthetic")(path)
    return get_wsi_reader(os.environ.get("WSI_READER", "openslide"))(path)


def get_slide_resolution(slide) -> tuple[float, float]:
    props = slide.properties
    if "openslide.mpp-x" in props and "openslide.mpp-y" in props:
        return float(props["openslide.mpp-x"]), float(props["openslide.mpp-y"])
    if "tiff.XResolution" in props and "tiff.YResolution" in props:
        x, y = float(props["tiff.XResolution"]), float(props["tiff.YResolution"])
        unit = props["tiff.ResolutionUnit"].lower()
        if unit == "centimeter":
            return 10000 / x, 10000 / y
        if unit == "inch":
            return 25400 / x, 25400 / y
        raise ValueError(f"Slide resolution unit not recognized: {props['tiff.ResolutionUnit']}")
    raise ValueError("Slide does not have MPP information")


def get_coords(tile_size: int, overlap: int, slide_dim: tuple[int, int], ts: float):
    """x outer / y inner, stride tile-overlap; a tile that would overrun the edge ends the
    row/column (the reference drops edge remainders, predict_wsi.py:385-391)."""
    out = []
    step = tile_size - overlap
    for i in range(0, slide_dim[0], step):
        if i + tile_size > slide_dim[0]:
            break
        for j in range(0, slide_dim[1], step):
            if j + tile_size > slide_dim[1]:
                break
            out.append(((int(i * ts), int(j * ts)), tile_size))
    return out


@dataclass
class SlidePlan:
    """Everything ``_init_slide`` derives before the first tile is read."""
    mpp: tuple[float, float]
    bounds: tuple[float, float]
    prediction_to_slide_scale: float
    level: int
    slide_dim: tuple[int, int]
    ts: float
    resize_factor: float
    read_tile_size: int
    read_overlap: int
    coords: list
    # level-0 pixels per prediction pixel, as main() derives it from the loader's shared mpp slots (predict_wsi.py:1517-1524,
    # 1793-1794).  Those slots are ``tmproc.Manager().Value("f", ...)`` = multiprocessing.managers.Value, which stores the
    # Python float as is (the typecode is ignored; only sharedctypes.Value would round to a C float) -- so mpp, bounds, ts and
    # resize_factor are all read back as full doubles and this equals prediction_to_slide_scale.
    polygon_scale: float = 1.0
    cell_tiles: object = None        # filled by run_rank: tile index (into coords) of every cell it returns
    rois: list | None = None
    tissue_cnts: list | None = None
    roi_class_dict: dict | None = None


def plan_slide(slide, tile_size: int, overlap: int, train_mpp: float) -> SlidePlan:
    mpp = get_slide_resolution(slide)
    bx = slide.properties.get("openslide.bounds-x")
    by = slide.properties.get("openslide.bounds-y")
    bounds = (float(bx) if bx is not None else 0.0, float(by) if by is not None else 0.0)
    scale = min(train_mpp / mpp[0], train_mpp / mpp[1])
    level = slide.get_best_level_for_downsample(scale)
    dim = tuple(slide.level_dimensions[level])
    ts = slide.level_downsamples[level]
    resize_factor = ts / scale
    read_tile = max(1, round(tile_size / resize_factor))
    read_ov = max(0, round(overlap / resize_factor))
    return SlidePlan(mpp, bounds, scale, level, dim, ts, resize_factor, read_tile, read_ov,
                     get_coords(read_tile, read_ov, dim, ts), polygon_scale=min(train_mpp / mpp[0], train_mpp / mpp[1]))


def read_tile(slide, plan: SlidePlan, coords) -> np.ndarray:
    """``fill_queue`` body for one tile up to the rescale: read_region -> drop alpha
    (uint8 h x w x 3 at the pyramid level's resolution).  The residual rescale to the
    model mpp (resize_tile_to_target_mpp) runs on the device after the H2D copy
    (``ops.resize_tile_to_target_mpp``)."""
    (x0, y0), ts = coords
    tile = np.array(slide.read_region((x0, y0), plan.level, (ts, ts)))
    if tile.shape[-1] == 4:
        tile = tile[:, :, :3]
    return np.ascontiguousarray(tile)
