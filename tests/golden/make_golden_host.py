"""Mint golden vectors for Classpose-owned HOST logic from the REFERENCE ITSELF (build container only):
``classpose.utils.get_slide_resolution`` (a1, utils.py:285-331) on fabricated property maps,
``predict_wsi.infer_structure`` (a5, predict_wsi.py:1377-1419) on synthetic checkpoints written to a temporary file,
``models.resolve_precision`` (a5, models.py:37-69) on the CPU device, imported under the stub finder of make_golden.py.
Fixtures hold inputs and outputs only.   python tests/golden/make_golden_host.py
"""
import json
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402


def main():
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, mg.REF)
    from classpose.utils import get_slide_resolution
    import classpose.entrypoints.predict_wsi as pw
    from classpose.models import resolve_precision
    from classpose_amd import synth

    res_cases = []
    for props in (
        {"openslide.mpp-x": "0.2521", "openslide.mpp-y": "0.2525"},
        {"openslide.mpp-x": 0.5, "openslide.mpp-y": 0.4999, "tiff.XResolution": "1", "tiff.YResolution": "1", "tiff.ResolutionUnit": "inch"},
        {"tiff.XResolution": "40000", "tiff.YResolution": "39990.5", "tiff.ResolutionUnit": "centimeter"},
        {"tiff.XResolution": "101600", "tiff.YResolution": 101000.0, "tiff.ResolutionUnit": "Inch"},
        {"tiff.XResolution": "5", "tiff.YResolution": "5", "tiff.ResolutionUnit": "furlong"},
        {"openslide.mpp-x": "0.25"},
        {},
    ):
        slide = types.SimpleNamespace(properties=props)
        try:
            out = {"mpp": list(get_slide_resolution(slide))}
        except Exception as e:                                   # the error type and text are part of the behaviour
            out = {"error": type(e).__name__, "message": str(e)}
        res_cases.append({"properties": props, **out})

    struct_cases = []
    for ncls, fts in ((7, None), (10, None), (7, [64, 128]), (3, [32])):
        sd = synth.make_state_dict(ncls, fts, depth=1, seed=3)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "m.pt")
            torch.save(sd, path)
            got_fts, got_ncls = pw.infer_structure(path)
        struct_cases.append({"n_classes_in": ncls, "fts_in": fts,
                             "keys": {k: list(v.shape) for k, v in sd.items()},
                             "fts": got_fts, "n_classes": got_ncls})

    prec_cases = []
    for p in ("fp32", "fp16", "bf16", "int8", "BF16"):
        try:
            prec_cases.append({"precision": p, "dtype": str(resolve_precision(p, torch.device("cpu")))})
        except Exception as e:
            prec_cases.append({"precision": p, "error": type(e).__name__, "message": str(e)})

    # a20 host helpers (grandqc/wsi_qc_helpers.py:26-150)
    import numpy as np
    from classpose.grandqc import wsi_qc_helpers as qh
    rng = np.random.default_rng(2)
    info_cases = []
    for dims, props, mpp_model in (((40000, 30000), {"openslide.mpp-x": "0.25", "openslide.mpp-y": "0.25"}, 1.0),
                                   ((98304, 65537), {"openslide.mpp-x": "0.2521", "openslide.mpp-y": "0.2521"}, 1.5),
                                   ((1000, 700), {"tiff.XResolution": "20000", "tiff.YResolution": "20000", "tiff.ResolutionUnit": "centimeter"}, 10.0)):
        slide = types.SimpleNamespace(level_dimensions=[dims], properties=props)
        w, h, mpp, td = qh.extract_slide_info(slide, mpp_model)
        info_cases.append({"dims": list(dims), "properties": props, "mpp_model": mpp_model,
                           "out": [w, h, mpp, list(td)]})
    feat_cases = []
    for n, closed in ((3, False), (4, False), (7, False), (6, True)):
        cnt = rng.integers(0, 500, (n, 2))
        if closed:
            cnt[-1] = cnt[0]
        scaling = np.array([3.9672, 4.0125])
        f = qh.create_geojson_feature(cnt, scaling, "Fold", [255, 99, 71])
        if f is not None:
            f["id"] = "uuid"
        feat_cases.append({"contour": cnt.tolist(), "scaling": scaling.tolist(), "feature": f})
    mask = rng.integers(0, 7, (9, 13))
    colors = [[int(v) for v in rng.integers(0, 256, 3)] for _ in range(7)]
    cmap_case = {"mask": mask.tolist(), "colors": colors, "rgb": qh.make_class_map(mask, colors).tolist()}

    # (b) drop-in boundary helpers: output file names and --device parsing (utils.py:29-148)
    from classpose import utils as ru
    name_cases = []
    for kind, base in (("cell_contours", "slide_1"), ("cell_centroids", "a.b"), ("tissue_contours", "x"),
                       ("artefact_contours", "x y"), ("roi", "r"), ("nonsense", "x")):
        try:
            name_cases.append({"kind": kind, "base": base, "name": ru.get_geojson_output_filename(kind, base)})
        except Exception as e:
            name_cases.append({"kind": kind, "base": base, "error": type(e).__name__, "message": str(e)})
    prefix_cases = [{"prefix": p_, "kind": k, "path": str(ru.get_geojson_output_path_from_prefix(p_, k))}
                    for p_, k in (("/out/dir/slide_1", "tissue_contours"), ("rel/s.v2", "artefact_contours"), ("plain", "roi"))]
    device_cases = [{"device": d, "devices": [str(x) for x in ru.get_device(d)]}
                    for d in ("cuda:0", "cuda:0,1,2,3", "cuda:3", "cpu", "cuda")]

    # --model_config surface (model_configs.py:20-148): the default table (paths relative to the model directory) and
    # the YAML schema
    import classpose.model_configs as rmc
    root = str(rmc.ROOT_MODEL_DIR)
    table = {k: dict(v, path=os.path.relpath(v["path"], root)) for k, v in rmc.DEFAULT_MODEL_CONFIGS.items()}
    yaml_cases = []
    for text in ("path: /w/my.pt\nmpp: 0.33\ncell_types: [A, B, C]\n",
                 "path: /w/my.pt\nmpp: 1\nurl: https://example.org/w.pt\nhf:\n  repo_id: org/repo\n  filename: w.pt\ncell_types:\n  - X\n"):
        with tempfile.TemporaryDirectory() as d:
            yp = os.path.join(d, "c.yaml")
            with open(yp, "w") as f:
                f.write(text)
            yaml_cases.append({"yaml": text, "config": rmc.ModelConfig.load_from_yaml(yp).model_dump()})

    # a1: SlideLoader._init_slide (predict_wsi.py:220-278) run unbound on a fake self whose shared values are REAL
    # multiprocessing.Manager().Value("f", ...) proxies -- what tmproc.Manager() hands the reference (predict_wsi.py:180-191):
    # managers.Value stores the Python object as is (the typecode is ignored; only sharedctypes.Value rounds to a C float),
    # so every slot reads back the full double -- and whose reader is a fabricated multi-level slide (OpenSlide's
    # level-selection rule)
    import multiprocessing
    manager = multiprocessing.Manager()
    init_cases = []
    for spec in (
        dict(dims=[(40000, 30000), (10000, 7500), (2500, 1875)], down=[1.0, 4.0, 16.0], props={"openslide.mpp-x": "0.25", "openslide.mpp-y": "0.25"},
             tile=1024, overlap=64, train_mpp=0.5),
        dict(dims=[(98304, 65536), (49152, 32768), (24573, 16382)], down=[1.0, 2.0, 4.000407], props={"openslide.mpp-x": "0.1213", "openslide.mpp-y": "0.1213"},
             tile=256, overlap=32, train_mpp=0.5),
        dict(dims=[(20000, 20000)], down=[1.0], props={"openslide.mpp-x": "0.2521", "openslide.mpp-y": "0.2498", "openslide.bounds-x": "15360.7", "openslide.bounds-y": "208.5"},
             tile=512, overlap=96, train_mpp=0.22),
        dict(dims=[(10000, 10000)], down=[1.0], props={"openslide.mpp-x": "0.5", "openslide.mpp-y": "0.5"}, tile=256, overlap=32, train_mpp=0.5),
        dict(dims=[(12000, 9000), (3000, 2250)], down=[1.0, 4.0], props={"tiff.XResolution": "20000", "tiff.YResolution": "20000", "tiff.ResolutionUnit": "centimeter"},
             tile=1024, overlap=64, train_mpp=0.25),
    ):
        class FakeSlide:
            properties = spec["props"]
            level_dimensions = [tuple(d) for d in spec["dims"]]
            level_downsamples = list(spec["down"])

            def get_best_level_for_downsample(self, ds):        # OpenSlide: the largest level whose downsample is <= ds
                best = 0
                for i, d in enumerate(self.level_downsamples):
                    if d <= ds:
                        best = i
                return best
        fv = lambda: manager.Value("f", 0)
        me = types.SimpleNamespace(slide_path="fake", tile_size=spec["tile"], overlap=spec["overlap"], train_mpp=spec["train_mpp"],
                                   roi_tree=None, mpp_x=fv(), mpp_y=fv(), bounds_x=fv(), bounds_y=fv(), ts=fv(), resize_factor=fv(),
                                   real_slide_path="fake")
        me.get_real_slide_path = lambda: "fake"
        me._get_coords = types.MethodType(pw.SlideLoader._get_coords, me)
        real_reader = pw.WSIReader
        pw.WSIReader = lambda path: FakeSlide()
        try:
            pw.SlideLoader._init_slide(me)
        finally:
            pw.WSIReader = real_reader
        init_cases.append({**spec, "mpp": list(me.mpp), "mpp_x": me.mpp_x.value, "mpp_y": me.mpp_y.value,
                           "bounds": [me.bounds_x.value, me.bounds_y.value], "level": me.level, "slide_dim": list(me.slide_dim),
                           "ts": me.ts.value, "resize_factor": me.resize_factor.value, "n_coords": len(me.coords),
                           "coords_head": [[list(map(int, c[0])), int(c[1])] for c in me.coords[:3]],
                           "coords_tail": [[list(map(int, c[0])), int(c[1])] for c in me.coords[-2:]],
                           "coords_sum": [int(sum(c[0][0] for c in me.coords)), int(sum(c[0][1] for c in me.coords))]})

    manager.shutdown()

    # a3: the output size of resize_tile_to_target_mpp (predict_wsi.py:102-123; Python's round-half-even on h * factor),
    # observed through a recording stand-in for cv2.resize
    import numpy as np
    seen = []
    pw.cv2.resize = lambda tile, size, interpolation=None: seen.append((size, interpolation)) or np.zeros((size[1], size[0], 3), np.uint8)
    pw.cv2.INTER_LINEAR = "LINEAR"
    resize_cases = []
    for h, w, rf in ((447, 447, 1.145909070968628), (251, 253, 0.5), (250, 255, 0.5), (264, 264, 0.97049880027771), (1, 3, 0.2),
                     (2048, 2048, 0.5), (512, 300, 2.0), (333, 777, 1.5), (256, 256, 1.0), (5, 5, 0.1)):
        seen.clear()
        out_tile = pw.resize_tile_to_target_mpp(np.zeros((h, w, 3), np.uint8), rf)
        resize_cases.append({"h": h, "w": w, "factor": rf, "out_hw": [int(out_tile.shape[0]), int(out_tile.shape[1])],
                             "called_resize": bool(seen), "interpolation": seen[0][1] if seen else None})

    # the command line itself (predict_wsi.py:1891-2021): what argparse hands to main() for a given argv
    cli_cases = []
    captured = []
    real_main, real_argv = pw.main, sys.argv
    pw.main = lambda a: captured.append(dict(vars(a)))
    try:
        for argv in (
            ["--model_config", "conic", "--slide_path", "s.svs", "--output_folder", "out"],
            ["--model_config", "cfg.yaml", "--slide_path", "/a/b.ndpi", "--output_folder", "/o", "--tta", "--filter_artefacts",
             "--tissue_detection_model_path", "t.pth", "--artefact_detection_model_path", "a.pth", "--roi_geojson", "r.geojson",
             "--roi_class_priority", "Tumour", "Stroma", "--min_area", "5000", "--batch_size", "32", "--device", "cuda:0,1",
             "--precision", "fp16", "--tile_size", "512", "--overlap", "96", "--output_type", "csv", "spatialdata",
             "--inference_threads", "3"],
            ["--model_config", "puma", "--slide_path", "s", "--output_folder", "o", "--no-tta", "--no-filter_artefacts",
             "--precision", "fp32"],
        ):
            sys.argv = ["classpose-predict-wsi"] + argv
            pw.main_with_args()
            cli_cases.append({"argv": argv, "args": captured[-1]})
    finally:
        pw.main, sys.argv = real_main, real_argv
    # the class-less sibling (predict_wsi_cpsam.py:677-812)
    import classpose.entrypoints.predict_wsi_cpsam as pc
    cpsam_cases = []
    real_main, real_argv = pc.main, sys.argv
    pc.main = lambda a: captured.append(dict(vars(a)))
    try:
        for argv in (
            ["--slide_path", "s.svs", "--output_folder", "out"],
            ["--model_path", "/w/cpsam_ft.pt", "--slide_path", "s", "--output_folder", "o", "--train_mpp", "0.25", "--tta",
             "--batch_size", "16", "--device", "cuda:1", "--tile_size", "512", "--precision", "fp32", "--overlap", "32",
             "--output_type", "csv", "--min_area", "100", "--roi_class_priority", "A"],
        ):
            sys.argv = ["classpose-predict-wsi-cpsam"] + argv
            pc.main_with_args()
            cpsam_cases.append({"argv": argv, "args": captured[-1]})
    finally:
        pc.main, sys.argv = real_main, real_argv

    with open(os.path.join(HERE, "reference_host.json"), "w") as f:
        json.dump({"resize_shape": resize_cases, "init_slide": init_cases, "cli": cli_cases, "cli_cpsam": cpsam_cases, "default_model_configs": table, "model_config_yaml": yaml_cases, "output_filename": name_cases, "output_path_from_prefix": prefix_cases, "get_device": device_cases,
                   "get_slide_resolution": res_cases, "infer_structure": struct_cases, "resolve_precision": prec_cases,
                   "extract_slide_info": info_cases, "create_geojson_feature": feat_cases, "make_class_map": cmap_case}, f, indent=1)
    print("wrote reference_host.json:", len(res_cases), len(struct_cases), len(prec_cases), len(info_cases), len(feat_cases))


if __name__ == "__main__":
    main()
