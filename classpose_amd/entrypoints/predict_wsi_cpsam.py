"""``classpose-predict-wsi-cpsam`` on the MI355X engine: the class-less Cellpose-SAM variant of the
WSI path (/root/reference/src/classpose/entrypoints/predict_wsi_cpsam.py:99-254, 665-816).  Same
tile loop, GrandQC / ROI filters, de-duplication and GeoJSON writers as ``predict_wsi``; the model
is a plain Cellpose-SAM checkpoint (``--model_path``, flows + cell probability only), the training
resolution comes from ``--train_mpp`` and every polygon is labelled "cell".
"""
from __future__ import annotations

import argparse

from . import predict_wsi

DEFAULT_TRAIN_MPP = 0.5


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Predict CellposeSAM cells and centroids for a whole-slide image (MI355X engine)")
    p.add_argument("--model_path", type=str, default="cpsam",
                   help="Path to the CellposeSAM model (a torch.save'd state dict)")
    p.add_argument("--slide_path", type=str, required=True)
    p.add_argument("--train_mpp", type=float, default=DEFAULT_TRAIN_MPP)
    p.add_argument("--tissue_detection_model_path", type=str, default=None)
    p.add_argument("--artefact_detection_model_path", type=str, default=None)
    p.add_argument("--filter_artefacts", action=argparse.BooleanOptionalAction, default=False)
    p.add_argument("--roi_geojson", type=str, default=None)
    p.add_argument("--roi_class_priority", type=str, nargs="+", default=None)
    p.add_argument("--min_area", type=int, default=0)
    p.add_argument("--tta", action=argparse.BooleanOptionalAction, default=False)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--tile_size", type=int, default=predict_wsi.DEFAULT_TILE_SIZE)
    p.add_argument("--precision", type=str, default="bf16", choices=["fp32", "fp16", "bf16"])
    p.add_argument("--overlap", type=int, default=predict_wsi.DEFAULT_OVERLAP)
    p.add_argument("--output_folder", type=str, required=True)
    p.add_argument("--output_type", type=str, nargs="+", default=None, choices=["csv", "spatialdata"])
    p.add_argument("--inference_threads", type=int, default=None,
                   help="accepted for compatibility; the engine overlaps stages with HIP streams instead")
    return p


def main(args, spawned: bool = False):
    args.model_config = None
    predict_wsi.main(args, spawned=spawned, parser_factory=build_parser)


def main_with_args():
    main(build_parser().parse_args())


if __name__ == "__main__":
    main_with_args()
