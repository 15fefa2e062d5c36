"""Stand-alone artefact detection CLI:
``python -m classpose_amd.grandqc.wsi_artefact_detection --slide_path S --output_path PREFIX
--model_art_path A --model_td_path T`` mirrors
/root/reference/src/classpose/grandqc/wsi_artefact_detection.py:351-420 -- writes
``PREFIX_artefact_mask.png``, ``PREFIX_artefact_map.png`` and the artefact GeoJSON.
"""
from __future__ import annotations

import argparse
import json

import numpy as np

from . import detect_artefacts_wsi
from ..entrypoints.predict_wsi import get_device
from ..wsi import WSIReader
from .wsi_tissue_detection import get_geojson_output_path_from_prefix


def main(argv=None):
    from PIL import Image
    p = argparse.ArgumentParser()
    p.add_argument("--slide_path", type=str, required=True)
    p.add_argument("--output_path", type=str, required=True)
    p.add_argument("--mpp_model_art", type=float, default=1.0, help="MPP of the artefact model")
    p.add_argument("--model_art_path", type=str, default="./models/artefact_detection/GrandQC_MPP1.pth")
    p.add_argument("--model_td_path", type=str, default="./models/tissue_detection/Tissue_Detection_MPP10.pth")
    p.add_argument("--min_area", type=int, default=0)
    p.add_argument("--device", type=str, default=None)
    args = p.parse_args(argv)
    device = get_device(args.device)[0]
    slide = WSIReader(args.slide_path)
    artefact_mask, artefact_map, _, geojson = detect_artefacts_wsi(
        slide, mpp_model_art=args.mpp_model_art, model_art_path=args.model_art_path, model_td_path=args.model_td_path,
        min_area=args.min_area, device=device, apply_bounds_offset=True)
    Image.fromarray(artefact_map).save(args.output_path + "_artefact_map.png")
    Image.fromarray(artefact_mask.astype(np.uint8)).save(args.output_path + "_artefact_mask.png")
    with open(get_geojson_output_path_from_prefix(args.output_path, "artefact_contours"), "w") as f:
        json.dump(geojson, f)


if __name__ == "__main__":
    main()
