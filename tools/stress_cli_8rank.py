"""The 8-ranks-on-one-GPU CLI run (tests/test_gpu_cli.py::test_predict_wsi_eight_ranks_share_one_gpu_gloo) repeated against ONE single-rank
run of the same slide: reports, for every repetition, whether the contour features are identical and -- if not -- what differs.
    python tools/stress_cli_8rank.py [repeats]"""
import json, os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(CLASSPOSE_SYNTHETIC_WEIGHTS="1", CLASSPOSE_SYNTHETIC_DEPTH="1", CLASSPOSE_AMD_PLUGINS="classpose_amd.synth:flow",
                  CPX_DIST_BACKEND="gloo")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tmp = tempfile.mkdtemp(prefix="cli8_")
os.environ["CLASSPOSE_MODEL_DIR"] = os.path.join(tmp, "nomodels")
from classpose_amd.entrypoints import predict_wsi


def args(out, device):
    d = {"model_config": "conic", "slide_path": "synthetic://1180x956?mpp=0.5&seed=53", "output_folder": out,
         "tissue_detection_model_path": None, "artefact_detection_model_path": None, "filter_artefacts": False,
         "roi_geojson": None, "roi_class_priority": None, "min_area": 0, "tta": True, "batch_size": 8,
         "device": device, "tile_size": 256, "precision": "bf16", "overlap": 32, "output_type": None, "inference_threads": 2}
    return type("Args", (), d)


def feats(out):
    fs = json.load(open([os.path.join(out, f) for f in os.listdir(out) if f.endswith("_cell_contours.geojson")][0]))["features"]
    return [(f["geometry"]["coordinates"], f["properties"]["classification"], f["properties"]["measurements"]) for f in fs]


def main():
    o1 = os.path.join(tmp, "one"); os.makedirs(o1)
    predict_wsi.main(args(o1, "cuda:0"))
    ref = feats(o1)
    bad = 0
    for r in range(reps):
        o8 = os.path.join(tmp, f"eight{r}"); os.makedirs(o8)
        predict_wsi.main(args(o8, "cuda:" + ",".join(["0"] * 8)))
        got = feats(o8)
        if got == ref:
            print(f"rep {r}: identical ({len(got)} cells)", flush=True)
        else:
            bad += 1
            first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a != b), None)
            print(f"rep {r}: DIFFERENT: {len(ref)} vs {len(got)} cells, first difference at {first}", flush=True)
            if first is not None:
                a, b = ref[first], got[first]
                print("   ref:", json.dumps(a[2])[:400]); print("   got:", json.dumps(b[2])[:400])
                print("   polygon equal:", a[0] == b[0], " class equal:", a[1] == b[1])
            sa, sb = {json.dumps(x, sort_keys=True) for x in ref}, {json.dumps(x, sort_keys=True) for x in got}
            print(f"   as sets: {len(sa - sb)} only in the 1-rank run, {len(sb - sa)} only in the 8-rank run", flush=True)
        shutil.rmtree(o8, ignore_errors=True)
    print("8-RANK STRESS", "clean" if bad == 0 else f"FAILED ({bad} of {reps})")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":        # (the CLI spawns its rank processes: the children re-import this module)
    main()
