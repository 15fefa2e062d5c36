"""Stand-alone tissue detection CLI (also spawned by the QuPath extension):
``python -m classpose_amd.grandqc.wsi_tissue_detection --slide_path S --output_path PREFIX``
mirrors /root/reference/src/classpose/grandqc/wsi_tissue_detection.py:332-387 -- writes
``PREFIX_image.png``, ``PREFIX_mask.png``, ``PREFIX_filled_class_map.png`` and the tissue GeoJSON.
"""
from __future__ import annotations

import argparse
import json

import numpy as np

from . import detect_tissue_wsi, draw_contour_outlines
from ..entrypoints.predict_wsi import get_device, get_geojson_output_filename
from ..wsi import WSIReader


def get_geojson_output_path_from_prefix(output_prefix: str, output_kind: str) -> str:
    """utils.get_geojson_output_path_from_prefix: <dir>/<template(basename)>"""
    import os
    d, base = os.path.split(output_prefix)
    return os.path.join(d, get_geojson_output_filename(output_kind, base))


def main(argv=None):
    from PIL import Image
    p = argparse.ArgumentParser()
    p.add_argument("--slide_path", type=str, required=True, help="Path to the slide")
    p.add_argument("--output_path", type=str, required=True, help="Path (prefix) of the outputs")
    p.add_argument("--model_path", type=str, default="./models/tissue_detection/Tissue_Detection_MPP10.pth")
    p.add_argument("--min_area", type=int, default=0, help="Minimum area of the polygon")
    p.add_argument("--device", type=str, default=None)
    args = p.parse_args(argv)
    device = get_device(args.device)[0]
    slide = WSIReader(args.slide_path)
    image, mask, _, cnts, geojson, _ = detect_tissue_wsi(slide, model_td_path=args.model_path, min_area=args.min_area,
                                                         device=device, apply_bounds_offset=True)
    Image.fromarray(image).save(args.output_path + "_image.png")
    Image.fromarray((mask * 255).astype(np.uint8)).save(args.output_path + "_mask.png")
    w_l0, h_l0 = slide.level_dimensions[0]
    scale = (w_l0 / mask.shape[1], h_l0 / mask.shape[0])
    bx = float(slide.properties.get("openslide.bounds-x", 0.0))
    by = float(slide.properties.get("openslide.bounds-y", 0.0))
    outlines = [(c["contour"] + np.array([bx, by])) / np.array(scale) for c in cnts.values()]
    Image.fromarray(draw_contour_outlines(mask.shape, outlines, 10)).save(args.output_path + "_filled_class_map.png")
    with open(get_geojson_output_path_from_prefix(args.output_path, "tissue_contours"), "w") as f:
        json.dump(geojson, f)


if __name__ == "__main__":
    main()
