"""``classpose-predict-wsi`` on the MI355X engine: same flags, same outputs.

Drop-in for /root/reference/src/classpose/entrypoints/predict_wsi.py (``main_with_args``
:1891-2019, ``main`` :1451-1886) on the tile path.  This module is the LIGHT front: the flags, the
flag checks and the parent of a multi-GPU run, none of which imports torch; the tile loop itself
(``TileStream``, ``run_rank``, ``write_outputs`` ...) lives in ``_tile_loop`` and is re-exported from
here on first use (module ``__getattr__``), so ``from classpose_amd.entrypoints.predict_wsi import
TileStream`` keeps working.

Differences by design (MI355X-first, results unchanged): tiles are batched across the slide
instead of one ``model.eval([tile])`` per tile; one process per GPU with STATIC sharding
(tile k -> rank k % world) instead of a shared queue; polygonisation runs in a thread pool of
the rank that produced the tile instead of one PostProcessor process for all devices.
Multi-GPU: either launch under ``torch.distributed.run`` or pass ``--device cuda:0,1,...``
(this process then spawns one worker per listed GPU, like the reference does,
predict_wsi.py:1542-1572 -- without importing torch itself: until round 5 the parent paid
1 - 1.6 s of imports before its children paid them again).

Also built: GrandQC tissue / artefact detection, ROI tile selection and cell filters,
``--output_type csv``, slides whose mpp differs from the model mpp, ``--precision fp32|fp16|bf16``.
Not built (raises, never silently ignored): ``--output_type spatialdata``.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import time

DEFAULT_TILE_SIZE = 1024
DEFAULT_OVERLAP = 64
MIN_TILE_SIZE = 256

GEOJSON_OUTPUT_TEMPLATES = {
    "cell_contours": os.getenv("CLASSPOSE_CELL_CONTOURS_GEOJSON", "{base_name}_cell_contours.geojson"),
    "cell_centroids": os.getenv("CLASSPOSE_CELL_CENTROIDS_GEOJSON", "{base_name}_cell_centroids.geojson"),
    "tissue_contours": os.getenv("CLASSPOSE_TISSUE_CONTOURS_GEOJSON", "{base_name}_tissue_contours.geojson"),
    "artefact_contours": os.getenv("CLASSPOSE_ARTEFACT_CONTOURS_GEOJSON", "{base_name}_artefact_contours.geojson"),
    "roi": os.getenv("CLASSPOSE_ROI_GEOJSON", "{base_name}_roi.geojson"),
}


def get_geojson_output_filename(output_kind: str, base_name: str) -> str:
    if output_kind not in GEOJSON_OUTPUT_TEMPLATES:
        raise ValueError(f"Invalid output kind: {output_kind}. Valid options are: "
                         + ", ".join(GEOJSON_OUTPUT_TEMPLATES))
    return GEOJSON_OUTPUT_TEMPLATES[output_kind].format(base_name=base_name)



def _check_unsupported(args):
    if args.output_type and "spatialdata" in args.output_type:
        raise NotImplementedError("--output_type spatialdata needs the spatialdata / geopandas stack, which is not "
                                  "available to this engine; csv is supported")
    if args.output_type and args.tissue_detection_model_path is None:
        raise ValueError(f"Tissue detection model path must be provided when using --output_type {args.output_type}")
    if args.tile_size < MIN_TILE_SIZE:
        raise ValueError(f"Tile size must be at least {MIN_TILE_SIZE}, got {args.tile_size}")


def _args_dict(args) -> dict:
    """Plain dict of an argparse namespace OR of the attribute-bag classes the reference's integration
    tests pass (``type("Args", (), {...})``, tests/test_prediction_integration.py:48-70)."""
    return {k: getattr(args, k) for k in dir(args) if not k.startswith("_")}



def _device_ids(device: str | None) -> list[int] | None:
    """``cuda:0,1,2`` -> [0, 1, 2] (utils.get_device of the reference, without torch); None when no index list is given."""
    if device is None or ":" not in device:
        return None
    return [int(i) for i in device.split(":")[1].split(",")]


def _spawn_entry(local_rank: int, world: int, port: int, arg_dict: dict, dev_ids: list[int]):
    os.environ.update(RANK=str(local_rank), WORLD_SIZE=str(world), LOCAL_RANK=str(dev_ids[local_rank]),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ..log import apply_rank_level
    apply_rank_level()                       # (a forked rank inherits loggers configured for rank 0)
    args = argparse.Namespace(**arg_dict)
    if not hasattr(args, "model_config"):
        args.model_config = None
    main(args, spawned=True)


def _start_method() -> str:
    """How the per-GPU workers are started.  ``spawn`` (a fresh interpreter per rank, what the reference does) costs every rank its own
    ``import torch``: 1 s alone, 2.9 s when eight start at once on a 16-core quota.  ``fork`` from a parent that has imported the tile loop but
    never touched a GPU gives every rank the imported modules for nothing -- allowed only while that holds (HIP does not survive a fork once it
    is initialised; nor do other threads' locks).  ``CLASSPOSE_START_METHOD=spawn|fork`` overrides the choice."""
    import threading
    forced = os.environ.get("CLASSPOSE_START_METHOD")
    if forced in ("spawn", "fork"):
        return forced
    if not sys.platform.startswith("linux") or threading.active_count() != 1:
        return "spawn"
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return "spawn"
    return "fork"


def _spawn_workers(args, ids: list[int]):
    """One child process per listed GPU (the reference starts its workers the same way, predict_wsi.py:1542-1572); they receive the parsed
    arguments, not sys.argv.  Plain ``multiprocessing``: forked from this process once it has imported the tile loop when that is safe
    (``_start_method``), freshly spawned otherwise; the first failing child ends the others and its exit code is reported."""
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    method = _start_method()
    try:
        import psutil
        os.environ.setdefault("CLASSPOSE_PARENT_T0", repr(psutil.Process().create_time()))
    except Exception:                                           # noqa: BLE001
        pass
    if method == "fork":
        t_i = time.time()
        from . import _tile_loop                                # noqa: F401 -- imported ONCE here, inherited by every rank
        if _start_method() != "fork":                           # (the import itself must not have started threads or initialised a GPU)
            method = "spawn"
        else:
            os.environ["CLASSPOSE_PARENT_IMPORT_S"] = f"{time.time() - t_i:.2f}"
    if method != "fork":
        os.environ.pop("CLASSPOSE_PARENT_IMPORT_S", None)
    ctx = mp.get_context(method)
    procs = [ctx.Process(target=_spawn_entry, args=(r, len(ids), port, _args_dict(args), ids), name=f"classpose-rank{r}")
             for r in range(len(ids))]
    for p in procs:
        p.start()
    failed = None
    try:
        while failed is None and any(p.is_alive() for p in procs):
            for p in procs:
                p.join(0.05)
                if p.exitcode not in (None, 0):
                    failed = p
                    break
        failed = failed or next((p for p in procs if p.exitcode not in (None, 0)), None)
    finally:
        if failed is not None or any(p.is_alive() for p in procs):
            for p in procs:
                if p.is_alive():
                    p.terminate()
            for p in procs:
                p.join(10)
    if failed is not None:
        raise RuntimeError(f"{failed.name} (pid {failed.pid}) exited with code {failed.exitcode}; the other ranks were stopped")


def main(args, spawned: bool = False, parser_factory=None):
    _check_unsupported(args)
    ids = _device_ids(args.device)
    if ids is not None and len(ids) > 1 and int(os.environ.get("WORLD_SIZE", 1)) == 1 and not spawned:
        _spawn_workers(args, ids)
        return
    from . import _tile_loop
    _tile_loop.run_main(args, spawned=spawned)


def build_parser() -> argparse.ArgumentParser:
    from ..model_configs import DEFAULT_MODEL_CONFIGS          # (pydantic + yaml: 0.2 s, only when a parser is built)
    p = argparse.ArgumentParser(description="Predict Classpose cells and centroids for a whole-slide image (MI355X engine)")
    p.add_argument("--model_config", type=str, required=True,
                   help="One of %s or a path to a YAML model config" % ", ".join(DEFAULT_MODEL_CONFIGS))
    p.add_argument("--slide_path", type=str, required=True)
    p.add_argument("--output_folder", type=str, required=True)
    p.add_argument("--tissue_detection_model_path", type=str, default=None)
    p.add_argument("--artefact_detection_model_path", type=str, default=None)
    p.add_argument("--filter_artefacts", action=argparse.BooleanOptionalAction, default=False)
    p.add_argument("--roi_geojson", type=str, default=None)
    p.add_argument("--roi_class_priority", type=str, nargs="+", default=None)
    p.add_argument("--min_area", type=int, default=0)
    p.add_argument("--tta", action=argparse.BooleanOptionalAction, default=False)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--precision", type=str, default="bf16", choices=["fp32", "fp16", "bf16"])
    p.add_argument("--tile_size", type=int, default=DEFAULT_TILE_SIZE)
    p.add_argument("--overlap", type=int, default=DEFAULT_OVERLAP)
    p.add_argument("--output_type", type=str, nargs="+", default=None, choices=["csv", "spatialdata"])
    p.add_argument("--inference_threads", type=int, default=None,
                   help="accepted for compatibility (the reference runs N Python threads per GPU to hide its "
                        "per-tile host work; here batches overlap on HIP streams and the value has no effect)")
    return p


def main_with_args():
    main(build_parser().parse_args())


def __getattr__(name: str):
    """Every other public name (TileStream, run_rank, CELL_ROW, get_device, write_outputs ...) is the tile loop's: imported on
    first use, so that importing this module -- all the parent of a multi-GPU run does -- stays free of torch."""
    if name.startswith("__"):
        raise AttributeError(name)
    mod = importlib.import_module("._tile_loop", __package__ or "classpose_amd.entrypoints")
    try:
        return getattr(mod, name)
    except AttributeError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None


if __name__ == "__main__":
    main_with_args()
