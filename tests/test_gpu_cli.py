"""GPU: the classpose-predict-wsi drop-in end to end on a small synthetic slide."""
import json
import os
import sys

import numpy as np
import pytest

from classpose_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_plugins():
    """The CLI runs in this process: plug-ins registered by one test must not leak into the next."""
    from classpose_amd import hooks
    hooks.reset()
    yield
    hooks.reset()


def test_predict_wsi_cli_flow_injection(cuda, tmp_path, monkeypatch):
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs = 1180, 956
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "conic", "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.5&seed=77",
        "--output_folder", str(tmp_path), "--tile_size", "256", "--overlap", "32", "--device", "cuda:0"])
    predict_wsi.main(args)
    files = sorted(p.name for p in tmp_path.glob("*.geojson"))
    assert len(files) == 2
    cont = json.load(open(tmp_path / [f for f in files if "contours" in f][0]))
    cent = json.load(open(tmp_path / [f for f in files if "centroids" in f][0]))
    assert cont["type"] == cent["type"] == "FeatureCollection"          # what the reference's test asserts
    assert len(cont["features"]) == len(cent["features"]) > 0
    f0 = cont["features"][0]
    assert f0["geometry"]["type"] == "Polygon" and f0["geometry"]["coordinates"][0][0] == f0["geometry"]["coordinates"][0][-1]
    assert [m["name"] for m in f0["properties"]["measurements"]] == ["area", "perimeter", "centroidX", "centroidY"]
    assert f0["properties"]["classification"]["name"] in ["Neutrophil", "Epithelial", "Lymphocyte", "Plasma cell",
                                                           "Eosinophil", "Connective"]
    # covered region: tiles at stride 224 that fit -> x < 5*224+... ; every nucleus fully inside
    # the covered area must be found exactly once (stitching + de-duplication), near its centre
    nx, ny = (W - 256) // 224 + 1, (Hs - 256) // 224 + 1
    cov_w, cov_h = (nx - 1) * 224 + 256, (ny - 1) * 224 + 256
    cx, cy, r, ident = synth.nuclei_in_region(77, 0, 0, cov_w, cov_h)
    inner = (cx - r > 12) & (cx + r < cov_w - 12) & (cy - r > 12) & (cy + r < cov_h - 12)
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[inner], cy[inner]], 1))
    assert np.all(d < 1.5), (d.max(), int((d >= 1.5).sum()))
    assert len(np.unique(idx)) == inner.sum()                          # one detection per nucleus
    counts = cKDTree(found).query_ball_point(np.stack([cx[inner], cy[inner]], 1), 7.5, return_length=True)
    assert np.all(counts == 1)                                         # duplicates from tile overlaps were removed
    # class label follows the nucleus id hash
    names = ["Neutrophil", "Epithelial", "Lymphocyte", "Plasma cell", "Eosinophil", "Connective"]
    exp = [names[int(i % np.uint64(6))] for i in ident[inner]]
    got = [cont["features"][j]["properties"]["classification"]["name"] for j in idx]
    assert got == exp


def test_classpose_model_eval_api(cuda):
    from classpose_amd.models import ClassposeModel
    sd = synth.make_state_dict(7, None, depth=1, seed=2)
    model = ClassposeModel(gpu=True, pretrained_model=sd, device=cuda, nclasses=7, precision="bf16")
    tile = synth.render_region(5, 0, 0, 256, 256)
    masks, flows, class_masks, styles = model.eval([tile, tile], batch_size=8, augment=False, bsize=256,
                                                   compute_masks=True)
    assert len(masks) == 2 and masks[0].dtype == np.uint16 and masks[0].shape == (256, 256)
    assert class_masks[0].dtype == np.int64 and class_masks[0].shape == (256, 256)
    assert flows[0][1].shape == (2, 256, 256) and flows[0][2].shape == (256, 256) and flows[0][3].shape == (7, 256, 256)
    assert np.array_equal(masks[0], masks[1])                           # deterministic
    m1, f1, c1, s1 = model.eval(tile)
    assert np.array_equal(m1, masks[0]) and np.array_equal(c1, class_masks[0])


def test_classpose_model_eval_parity_and_concurrency(cuda):
    """ClassposeModel.eval (models.py:478-827 of the reference) against the oracle, for one tile and for a list: flows /
    cellprob / class logits within the half-precision tolerance of the float32 oracle network, the id map bit-exact w.r.t.
    the oracle dynamics on the RETURNED fields, class_masks == the oracle's class vote -- and two Python threads calling
    eval on ONE model at the same time (the reference's --inference_threads, predict_wsi.py:790-797) get exactly what the
    serial calls return (engines come from a bounded pool on the shared weights; no lock around the GPU work) -- and six
    more short-lived threads afterwards create no further engine (the pool is capped at max_engines, callers beyond it queue)."""
    import threading
    from classpose_amd.models import ClassposeModel
    from oracle import classmask, dynamics, net as onet, tiling
    sd = synth.make_state_dict(7, None, depth=2, seed=12)
    model = ClassposeModel(gpu=True, pretrained_model=sd, device=cuda, nclasses=7, precision="bf16", max_engines=2)
    tiles = [synth.render_region(77, 0, 0, 256, 256), synth.render_region(77, 300, 120, 256, 256),
             synth.render_region(78, 40, 700, 256, 256)]
    masks, flows, class_masks, styles = model.eval(tiles, batch_size=8, augment=False, bsize=256, compute_masks=True)
    fw = onet.make_forward(sd)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    for i, t in enumerate(tiles):
        x = tiling.normalize_img(t[None])
        dP, cp, yc = tiling.run_net(fw, x, batch_size=8)
        _, dP_i, cp_i, yc_i, shp = flows[i]
        assert rel(dP_i, dP) < 2e-2 and rel(cp_i, cp) < 2e-2 and rel(yc_i, yc) < 2e-2
        ref = dynamics.compute_masks(dP_i, cp_i)
        assert masks[i].dtype == np.uint16 and np.array_equal(masks[i], ref)
        cm, _ = classmask.compute_class_masks(ref, yc_i)
        assert class_masks[i].dtype == np.int64 and np.array_equal(class_masks[i], cm)
        assert shp == (1, 256, 256, 3) and styles[i].shape == (256,)
    m1, f1, c1, _ = model.eval(tiles[1])                                  # a single array: the un-listed return form
    assert np.array_equal(m1, masks[1]) and np.array_equal(c1, class_masks[1]) and np.array_equal(f1[1], flows[1][1])
    # two threads, one model, many calls each
    res, err = {}, []
    def work(tid, order):
        try:
            out = []
            for _ in range(4):
                for j in order:
                    m, f, c, _s = model.eval(tiles[j])
                    out.append((j, m, f[1], f[2], c))
            res[tid] = out
        except Exception as e:                                            # pragma: no cover
            err.append(e)
    th = [threading.Thread(target=work, args=(0, [0, 1, 2])), threading.Thread(target=work, args=(1, [2, 0, 1]))]
    for t in th: t.start()
    for t in th: t.join()
    assert not err, err
    assert 1 <= model.engines_alive() <= 2
    # a caller that starts a fresh thread per request: six more threads, three at a time in flight -> still at most max_engines engines
    th = [threading.Thread(target=work, args=(k, [k % 3])) for k in range(2, 8)]
    for t in th: t.start()
    for t in th: t.join()
    assert not err, err
    assert model.engines_alive() <= 2
    for tid in range(8):
        for j, m, dP_t, cp_t, c in res[tid]:
            assert np.array_equal(m, masks[j]) and np.array_equal(c, class_masks[j])
            assert np.array_equal(dP_t, flows[j][1]) and np.array_equal(cp_t, flows[j][2])


def test_predict_wsi_cli_default_tile_1024(cuda, tmp_path, monkeypatch):
    """the reference's DEFAULT geometry: --tile_size 1024 --overlap 64 (25 sub-tiles per tile)"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs = 2100, 1100                                   # 2 x 1 tiles at stride 960
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "puma", "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.22&seed=5",
        "--output_folder", str(tmp_path), "--device", "cuda:0"])
    assert args.tile_size == 1024 and args.overlap == 64
    predict_wsi.main(args)
    cont = json.load(open(next(tmp_path.glob("*contours.geojson"))))
    cx, cy, r, ident = synth.nuclei_in_region(5, 0, 0, 1984, 1024)
    inner = (cx - r > 12) & (cx + r < 1984 - 12) & (cy - r > 12) & (cy + r < 1024 - 12)
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[inner], cy[inner]], 1))
    assert np.all(d < 1.5) and len(np.unique(idx)) == inner.sum()
    names = ["Apoptosis", "Tumor", "Endothelial", "Stroma", "Lymphocyte", "Histocyte", "Epithelial", "Melanophage", "Other"]
    assert [cont["features"][j]["properties"]["classification"]["name"] for j in idx] == \
        [names[int(i % np.uint64(9))] for i in ident[inner]]


def test_predict_wsi_cli_rescaled_slide(cuda, tmp_path, monkeypatch):
    """slide mpp 0.4 vs model mpp 0.5: 320 px reads are rescaled to 256 px on the device
    (resize_tile_to_target_mpp) and the polygons come back in level-0 coordinates"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    from classpose_amd import wsi
    W, Hs = 1000, 700
    uri = f"synthetic://{W}x{Hs}?mpp=0.4&seed=9"
    plan = wsi.plan_slide(wsi.WSIReader(uri), 256, 32, 0.5)
    assert plan.read_tile_size == 320 and plan.read_overlap == 40 and abs(plan.resize_factor - 0.8) < 1e-6
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "conic", "--slide_path", uri, "--output_folder", str(tmp_path),
        "--tile_size", "256", "--overlap", "32", "--device", "cuda:0"])
    predict_wsi.main(args)
    cont = json.load(open(next(tmp_path.glob("*contours.geojson"))))
    nx, ny = (W - 320) // 280 + 1, (Hs - 320) // 280 + 1
    cov_w, cov_h = (nx - 1) * 280 + 320, (ny - 1) * 280 + 320
    cx, cy, r, ident = synth.nuclei_in_region(9, 0, 0, cov_w, cov_h)
    inner = (cx - r > 15) & (cx + r < cov_w - 15) & (cy - r > 15) & (cy + r < cov_h - 15)
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    areas = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"] == "area"][0]
                      for f in cont["features"]])
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[inner], cy[inner]], 1))
    # 4-7 px radii at the network's resolution: the dynamics may drop the odd tiny nucleus
    # (same arithmetic as the reference); everything found sits on its nucleus, once
    ok = d < 2.0
    assert ok.mean() > 0.99 and np.median(d) < 0.5, (d.max(), int((~ok).sum()))
    assert len(np.unique(idx[ok])) == ok.sum()
    idx, r_in = idx[ok], r[inner][ok]
    # areas are reported in level-0 pixels: within the discretisation error of pi r^2
    rel = np.abs(areas[idx] - np.pi * r_in ** 2) / (np.pi * r_in ** 2)
    # (vertices run through boundary pixel centres: ~1/r of the disc is lost; unscaled areas would be off by 0.46)
    assert np.median(rel) < 0.3, np.median(rel)


def test_predict_wsi_cli_roi(cuda, tmp_path, monkeypatch):
    """--roi_geojson: ROI-driven tile grid (300 px tiles here), tiles outside skipped, cells kept
    only when their centroid is inside an ROI"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    rois = [[[400, 300], [1000, 300], [1000, 600], [400, 600], [400, 300]],                 # 600 x 300
            [[1200, 900], [1500, 900], [1500, 1300], [1350, 1400], [1200, 1300], [1200, 900]]]
    roi_path = tmp_path / "roi.geojson"
    roi_path.write_text(json.dumps({"type": "FeatureCollection", "features": [
        {"type": "Feature", "geometry": {"type": "Polygon", "coordinates": [r]}, "properties": {}} for r in rois]}))
    out = tmp_path / "out"
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "conic", "--slide_path", "synthetic://2000x1600?mpp=0.5&seed=21",
        "--output_folder", str(out), "--tile_size", "512", "--overlap", "64", "--device", "cuda:0",
        "--roi_geojson", str(roi_path)])
    predict_wsi.main(args)
    cont = json.load(open(next(out.glob("*contours.geojson"))))
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    from classpose_amd import roi
    polys = [roi.RoiPolygon(r) for r in rois]
    assert all(any(p.contains_point_strict(x, y) for p in polys) for x, y in found)
    cx, cy, r, ident = synth.nuclei_in_region(21, 0, 0, 2000, 1600)
    # nuclei whose whole disc is inside an ROI must all be there, once
    def disc_inside(p, x, y, rad):
        return all(p.contains_point_strict(x + dx * (rad + 1.5), y + dy * (rad + 1.5))
                   for dx, dy in [(1, 0), (-1, 0), (0, 1), (0, -1), (.7, .7), (-.7, .7), (.7, -.7), (-.7, -.7)])
    inner = np.array([any(disc_inside(p, x, y, rr) for p in polys) for x, y, rr in zip(cx, cy, r)])
    assert inner.sum() > 150
    # the pulled-back last row/column of an ROI grid overlaps its neighbour heavily, so a nucleus
    # can be detected 4-6 times with identical area; the reference's greedy de-duplication
    # (predict_wsi.py:931-960, mirrored verbatim) can then drop EVERY copy.  Nuclei seen by at most
    # three tiles never hit that quirk and must be there exactly once.
    tiles = roi.get_coords_roi(polys, 512, 64, 64, 1.0)
    copies = sum(((cx + r >= x0) & (cx - r <= x0 + s) & (cy + r >= y0) & (cy - r <= y0 + s)).astype(int)
                 for (x0, y0), s in tiles)
    safe = inner & (copies <= 3)
    assert safe.sum() > 100
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[safe], cy[safe]], 1))
    assert np.all(d < 1.5) and len(np.unique(idx)) == safe.sum()
    counts = cKDTree(found).query_ball_point(np.stack([cx[inner], cy[inner]], 1), 7.5, return_length=True)
    assert np.all(counts <= 1) and (counts == 1).mean() > 0.9


def test_predict_wsi_cli_tissue_and_artefacts(cuda, tmp_path, monkeypatch):
    """--tissue_detection_model_path / --artefact_detection_model_path / --filter_artefacts: GrandQC nets
    run on the thumbnails (synthetic weights), class maps injected analytically; tiles outside tissue are
    skipped, cells are filtered by tissue and artefact polygons, both contour files are written"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow+qc")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs = 3000, 2400
    out = tmp_path / "out"
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "conic", "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.5&seed=31",
        "--output_folder", str(out), "--tile_size", "256", "--overlap", "32", "--device", "cuda:0",
        "--tissue_detection_model_path", str(tmp_path / "td.pth"),
        "--artefact_detection_model_path", str(tmp_path / "art.pth"), "--filter_artefacts",
        "--output_type", "csv"])
    predict_wsi.main(args)
    import pandas as pd
    df = pd.read_csv(next(out.glob("*_cell_densities.csv")))
    assert list(df.columns) == ["region", "cell_class", "count", "density"] and len(df) == 6
    assert set(df["region"]) == {"tissue"}
    cont = json.load(open(next(out.glob("*cell_contours.geojson"))))
    tissue = json.load(open(next(out.glob("*tissue_contours.geojson"))))
    art = json.load(open(next(out.glob("*artefact_contours.geojson"))))
    assert df["count"].sum() == len(cont["features"])
    assert len(tissue["features"]) == 1 and len(tissue["features"][0]["geometry"]["coordinates"]) == 2   # ellipse + hole
    assert tissue["features"][0]["id"] == "tissue_0"
    assert [m["name"] for m in tissue["features"][0]["properties"]["measurements"]] == ["area", "perimeter", "centroidX", "centroidY"]
    assert len(art["features"]) == 1 and art["features"][0]["properties"]["classification"]["name"] == "artefact"
    t_area = tissue["features"][0]["properties"]["measurements"][0]["value"]
    a_area = art["features"][0]["properties"]["measurements"][0]["value"]
    eff_mm2 = (t_area - a_area) * 0.25 / 1e6
    assert np.allclose(df["density"], df["count"] / eff_mm2)
    ell_area = np.pi * 0.40 * W * 0.36 * Hs - np.pi * (0.06 * W) ** 2
    assert abs(t_area - ell_area) / ell_area < 0.05
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    u, v = found[:, 0] / W, found[:, 1] / Hs

    def in_tissue(u, v, grow):
        ell = ((u - 0.5) / (0.40 + grow)) ** 2 + ((v - 0.5) / (0.36 + grow)) ** 2 <= 1.0
        hole = (u - 0.62) ** 2 + ((v - 0.55) * Hs / W) ** 2 <= (0.06 - grow) ** 2
        return ell & ~hole

    def in_fold(u, v, grow):
        return (u > 0.30 - grow) & (u < 0.42 + grow) & (v > 0.30 - grow) & (v < 0.45 + grow)
    tol = 1.5 * 20 / W                                       # 1.5 thumbnail pixels of the 10 um/px tissue map
    assert np.all(in_tissue(u, v, tol)) and not np.any(in_fold(u, v, -tol))
    cx, cy, r, ident = synth.nuclei_in_region(31, 0, 0, W, Hs)
    safe = in_tissue(cx / W, cy / Hs, -3 * tol) & ~in_fold(cx / W, cy / Hs, 3 * tol) & \
        (cx > 300) & (cx < W - 300) & (cy > 300) & (cy < Hs - 300)
    assert safe.sum() > 2000
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[safe], cy[safe]], 1))
    # (a nucleus in a 4-tile corner overlap can lose every copy to the reference's greedy de-duplication)
    assert (d < 1.5).mean() > 0.995 and len(np.unique(idx[d < 1.5])) == (d < 1.5).sum()


def test_predict_wsi_cpsam_cli(cuda, tmp_path, monkeypatch):
    """class-less Cellpose-SAM entry point: --model_path / --train_mpp, every polygon is a "cell" """
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    from classpose_amd.entrypoints import predict_wsi_cpsam
    W, Hs = 800, 600
    args = predict_wsi_cpsam.build_parser().parse_args([
        "--model_path", str(tmp_path / "cpsam"), "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.5&seed=3",
        "--output_folder", str(tmp_path), "--tile_size", "256", "--overlap", "32", "--device", "cuda:0"])
    predict_wsi_cpsam.main(args)
    cont = json.load(open(next(tmp_path.glob("*contours.geojson"))))
    assert len(cont["features"]) > 100
    assert {f["properties"]["classification"]["name"] for f in cont["features"]} == {"cell"}
    assert cont["features"][0]["properties"]["classification"]["color"] == [0, 168, 132]
    nx, ny = (W - 256) // 224 + 1, (Hs - 256) // 224 + 1
    cov_w, cov_h = (nx - 1) * 224 + 256, (ny - 1) * 224 + 256
    cx, cy, r, _ = synth.nuclei_in_region(3, 0, 0, cov_w, cov_h)
    inner = (cx - r > 12) & (cx + r < cov_w - 12) & (cy - r > 12) & (cy + r < cov_h - 12)
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[inner], cy[inner]], 1))
    assert (d < 1.5).mean() > 0.99


def _reference_integration_args(slide, out, **over):
    """the attribute bag of the reference's integration tests
    (/root/reference/tests/test_prediction_integration.py:48-70): conic, 256 / 64, fp32, batch 1"""
    d = {"model_config": "conic", "slide_path": slide, "output_folder": str(out),
         "tissue_detection_model_path": None, "artefact_detection_model_path": None, "filter_artefacts": False,
         "roi_geojson": None, "roi_class_priority": None, "min_area": 0, "tta": False, "batch_size": 1,
         "device": "cuda", "tile_size": 256, "precision": "fp32", "overlap": 64, "output_type": None,
         "inference_threads": 2}
    d.update(over)
    return type("Args", (), d)


@pytest.mark.parametrize("threads", [2, 1])
def test_predict_wsi_reference_integration_args_fp32(cuda, tmp_path, monkeypatch, threads):
    """test_predict_wsi_integration / ..._one_inference_thread of the reference, same argument set
    (--precision fp32 through the exact-f32 MFMA kernels), same assertions + the nuclei check"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "2")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs = 840, 650
    predict_wsi.main(_reference_integration_args(f"synthetic://{W}x{Hs}?mpp=0.5&seed=41", tmp_path,
                                                 inference_threads=threads))
    cont_p, cent_p = next(tmp_path.glob("*_cell_contours.geojson")), next(tmp_path.glob("*_cell_centroids.geojson"))
    assert cont_p.exists() and cent_p.exists()
    cont = json.load(open(cont_p))
    assert cont["type"] == "FeatureCollection"
    nx, ny = (W - 256) // 192 + 1, (Hs - 256) // 192 + 1
    cov_w, cov_h = (nx - 1) * 192 + 256, (ny - 1) * 192 + 256
    cx, cy, r, _ = synth.nuclei_in_region(41, 0, 0, cov_w, cov_h)
    inner = (cx - r > 12) & (cx + r < cov_w - 12) & (cy - r > 12) & (cy + r < cov_h - 12)
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[inner], cy[inner]], 1))
    assert (d < 1.5).mean() > 0.99 and len(np.unique(idx[d < 1.5])) == (d < 1.5).sum()


def test_predict_wsi_cpsam_reference_integration_args_fp32(cuda, tmp_path, monkeypatch):
    """test_predict_wsi_cpsam_integration of the reference (:172-215)"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    from classpose_amd.entrypoints import predict_wsi_cpsam
    args = _reference_integration_args("synthetic://700x520?mpp=0.5&seed=43", tmp_path, model_path="cpsam", train_mpp=0.5)
    predict_wsi_cpsam.main(args)
    assert next(tmp_path.glob("*_cell_contours.geojson")).exists() and next(tmp_path.glob("*_cell_centroids.geojson")).exists()


def test_predict_wsi_multi_gpu_integration(tmp_path, monkeypatch):
    """test_predict_wsi_multi_gpu_integration of the reference (:131-170): --device cuda:0,1 spawns one fresh
    worker process per GPU (static tile shard k % 2, RCCL all-gather of the cell tables); the result must equal
    the single-GPU run cell for cell."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("Needs at least 2 GPUs")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    slide = "synthetic://1500x1100?mpp=0.5&seed=45"
    o1, o2 = tmp_path / "one", tmp_path / "two"
    predict_wsi.main(_reference_integration_args(slide, o1, device="cuda:0", precision="bf16"))
    predict_wsi.main(_reference_integration_args(slide, o2, device="cuda:0,1", precision="bf16"))
    key = lambda f: tuple(m["value"] for m in f["properties"]["measurements"])
    a = sorted(key(f) for f in json.load(open(next(o1.glob("*_cell_contours.geojson"))))["features"])
    b = sorted(key(f) for f in json.load(open(next(o2.glob("*_cell_contours.geojson"))))["features"])
    assert len(a) > 100 and a == b


def test_predict_wsi_cli_puma_grandqc(cuda, tmp_path, monkeypatch):
    """BASELINE configs[2] in one run (scaled down): puma (10 classes, 0.22 um/px, default 1024 / 64 tiles = 25
    sub-tiles) WITH GrandQC tissue + artefact detection, --filter_artefacts and --output_type csv"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow+qc")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs = 6800, 5200
    out = tmp_path / "out"
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "puma", "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.22&seed=47",
        "--output_folder", str(out), "--device", "cuda:0",
        "--tissue_detection_model_path", str(tmp_path / "td.pth"),
        "--artefact_detection_model_path", str(tmp_path / "art.pth"), "--filter_artefacts",
        "--output_type", "csv"])
    assert args.tile_size == 1024 and args.overlap == 64 and args.precision == "bf16"
    predict_wsi.main(args)
    import pandas as pd
    df = pd.read_csv(next(out.glob("*_cell_densities.csv")))
    names = ["Apoptosis", "Tumor", "Endothelial", "Stroma", "Lymphocyte", "Histocyte", "Epithelial", "Melanophage", "Other"]
    assert list(df["cell_class"]) == names and set(df["region"]) == {"tissue"}
    cont = json.load(open(next(out.glob("*cell_contours.geojson"))))
    tissue = json.load(open(next(out.glob("*tissue_contours.geojson"))))
    art = json.load(open(next(out.glob("*artefact_contours.geojson"))))
    assert df["count"].sum() == len(cont["features"]) > 500
    assert len(tissue["features"]) == 1 and len(art["features"]) == 1
    t_area = tissue["features"][0]["properties"]["measurements"][0]["value"]
    a_area = art["features"][0]["properties"]["measurements"][0]["value"]
    assert np.allclose(df["density"], df["count"] / ((t_area - a_area) * 0.22 * 0.22 / 1e6))
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    u, v = found[:, 0] / W, found[:, 1] / Hs
    tol = 1.5 * (10 / 0.22) / W
    ell = ((u - 0.5) / (0.40 + tol)) ** 2 + ((v - 0.5) / (0.36 + tol)) ** 2 <= 1.0
    fold = (u > 0.30 + tol) & (u < 0.42 - tol) & (v > 0.30 + tol) & (v < 0.45 - tol)
    assert np.all(ell) and not np.any(fold)
    # every class label follows the nucleus id hash (9 cell types)
    cx, cy, r, ident = synth.nuclei_in_region(47, 0, 0, W, Hs)
    from scipy.spatial import cKDTree
    d, idx = cKDTree(np.stack([cx, cy], 1)).query(found)
    assert (d < 1.5).mean() > 0.995
    got = [f["properties"]["classification"]["name"] for f in cont["features"]]
    ok = d < 1.5
    assert [g for g, k in zip(got, ok) if k] == [names[int(i % np.uint64(9))] for i in ident[idx[ok]]]


def test_predict_wsi_cli_roi_class_densities_with_artefacts(cuda, tmp_path, monkeypatch):
    """--roi_geojson with classified ROIs + tissue + artefact detection + --output_type csv: densities per ROI
    class over (ROI area - artefact area inside the ROI), the artefact share from polygon intersection areas
    (predict_wsi.py:1797-1836)"""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow+qc")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs = 3000, 2400
    fold = (0.30 * W, 0.30 * Hs, 0.42 * W, 0.45 * Hs)                       # the analytic artefact rectangle
    rect = lambda x0, y0, x1, y1: [[x0, y0], [x1, y0], [x1, y1], [x0, y1], [x0, y0]]
    rois = {"tumour": rect(1000, 700, 1700, 1300),                          # overlaps the fold partially
            "stroma": rect(1800, 1400, 2300, 1900)}                          # no artefact inside
    roi_path = tmp_path / "roi.geojson"
    roi_path.write_text(json.dumps({"type": "FeatureCollection", "features": [
        {"type": "Feature", "geometry": {"type": "Polygon", "coordinates": [r]},
         "properties": {"classification": {"name": k}}} for k, r in rois.items()]}))
    out = tmp_path / "out"
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "conic", "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.5&seed=33",
        "--output_folder", str(out), "--tile_size", "256", "--overlap", "32", "--device", "cuda:0",
        "--roi_geojson", str(roi_path), "--tissue_detection_model_path", str(tmp_path / "td.pth"),
        "--artefact_detection_model_path", str(tmp_path / "art.pth"), "--output_type", "csv"])
    predict_wsi.main(args)
    import pandas as pd
    df = pd.read_csv(next(out.glob("*_cell_densities.csv")))
    assert set(df["region"]) == {"tumour", "stroma"} and len(df) == 12
    ix0, iy0 = max(fold[0], 1000), max(fold[1], 700)
    ix1, iy1 = min(fold[2], 1700), min(fold[3], 1300)
    inter = (ix1 - ix0) * (iy1 - iy0)
    assert inter > 50000
    for name, r in rois.items():
        area = (r[2][0] - r[0][0]) * (r[2][1] - r[0][1])
        eff = area - (inter if name == "tumour" else 0.0)
        d = df[df["region"] == name]
        got_eff = (d["count"] / d["density"]).replace([np.inf, -np.inf], np.nan).dropna() * 1e6 / 0.25    # pixels
        assert len(got_eff) > 0
        # the artefact contour follows 1.5 um/px thumbnail pixels (3 level-0 px): a few 1e-3 of the ROI area
        assert np.allclose(got_eff, eff, rtol=0.01), (name, float(got_eff.iloc[0]), eff)


@pytest.mark.parametrize("props_kind", ["openslide-mpp+bounds", "tiff-resolution"])
def test_predict_wsi_cli_openslide_protocol_reader_at_level_1(cuda, tmp_path, monkeypatch, props_kind):
    """f4: an OpenSlide-PROTOCOL reader through the tile loop (predict_wsi.py:220-278, 446-451, 463-482; reader switch
    src/classpose/__init__.py:6-41): ``WSI_READER=openslide`` imports ``openslide.OpenSlide`` (tests/fake_openslide.py
    installed as that module), a 3-level pyramid, mpp from ``openslide.mpp-*`` or from ``tiff.*Resolution``, a level > 0
    chosen by ``get_best_level_for_downsample``, ``read_region`` in level-0 coordinates returning RGBA ``PIL.Image``s, a
    residual rescale (0.8) on the device, and ``openslide.bounds-*`` subtracted from the outputs.  The raw cells of the
    tile loop equal those of the ``synthetic://`` run of the same pixels under the exact level-1 -> level-0 map (x2), and the
    files of the full CLI run are those cells after the reference's de-duplication and bounds shift."""
    import sys
    import fake_openslide as fo
    from classpose_amd import geojson, hooks, wsi
    from classpose_amd.entrypoints import predict_wsi
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "fake_openslide")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    monkeypatch.setenv("WSI_READER", "openslide")
    monkeypatch.setitem(sys.modules, "openslide", fo)
    W1, H1, seed = 1000, 700, 9
    if props_kind == "tiff-resolution":
        props, bounds = {"tiff.XResolution": "50000", "tiff.YResolution": "50000", "tiff.ResolutionUnit": "centimeter"}, (0.0, 0.0)
    else:
        props, bounds = {"openslide.mpp-x": "0.2", "openslide.mpp-y": "0.2", "openslide.bounds-x": "100", "openslide.bounds-y": "60.5"}, (100.0, 60.5)
    path = str(tmp_path / "fake_slide.svs")
    fo.SLIDES[path] = dict(seed=seed, base_level=1, base_dims=(W1, H1), downsamples=[1.0, 2.0, 8.0], properties=props)
    fo.READS.clear()
    out_f, out_s = tmp_path / "fake", tmp_path / "synth"
    common = ["--model_config", "conic", "--tile_size", "256", "--overlap", "32", "--device", "cuda:0"]
    args_f = predict_wsi.build_parser().parse_args(common + ["--slide_path", path, "--output_folder", str(out_f)])
    args_s = predict_wsi.build_parser().parse_args(common + ["--slide_path", f"synthetic://{W1}x{H1}?mpp=0.4&seed={seed}",
                                                             "--output_folder", str(out_s)])
    # ---- slide planning: level 1 (downsample 2 <= 0.5 / 0.2 = 2.5), 320-px reads, residual factor 0.8
    slide = wsi.WSIReader(path)
    assert isinstance(slide, fo.OpenSlide) and slide.level_dimensions == ((2000, 1400), (1000, 700), (250, 175))
    plan = wsi.plan_slide(slide, 256, 32, 0.5)
    assert plan.level == 1 and plan.ts == 2.0 and plan.resize_factor == 0.8 and plan.read_tile_size == 320
    assert plan.mpp == (0.2, 0.2) and plan.bounds == bounds and plan.polygon_scale == 2.5
    assert len(plan.coords) == 6 and plan.coords[1] == ((0, 560), 320) and plan.coords[-1] == ((1120, 560), 320)
    # ---- the tile loop on both readers
    cells_f, xy_f, labels, _ = predict_wsi.run_rank(args_f, 0, 1, cuda)
    assert len(fo.READS) == 6 and all(r[2] == 1 and r[3] == (320, 320) for r in fo.READS)         # every read at level 1
    assert sorted(r[1] for r in fo.READS) == sorted(c[0] for c in plan.coords)                   # origins in level-0 pixels
    hooks.reset()
    cells_s, xy_s, _, plan_s = predict_wsi.run_rank(args_s, 0, 1, cuda)
    assert plan_s.level == 0 and plan_s.resize_factor == 0.8 and plan_s.polygon_scale == 1.25
    assert len(cells_f) == len(cells_s) > 150
    assert np.array_equal(xy_f, 2.0 * xy_s)                                    # same contours, level-1 -> level-0 pixels
    assert np.array_equal(cells_f["area"], 4.0 * cells_s["area"]) and np.array_equal(cells_f["perimeter"], 2.0 * cells_s["perimeter"])
    assert np.array_equal(cells_f["cx"], 2.0 * cells_s["cx"]) and np.array_equal(cells_f["cy"], 2.0 * cells_s["cy"])
    assert np.array_equal(cells_f["cls"], cells_s["cls"]) and np.array_equal(cells_f["n_pts"], cells_s["n_pts"])
    # every nucleus well inside the covered area is found, at 2x its level-1 position
    cx, cy, r, _ = synth.nuclei_in_region(seed, 0, 0, 880, 600)
    inner = (cx - r > 15) & (cx + r < 880 - 15) & (cy - r > 15) & (cy + r < 600 - 15)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(np.stack([cells_f["cx"], cells_f["cy"]], 1)).query(2.0 * np.stack([cx[inner], cy[inner]], 1))
    assert (d < 4.0).mean() > 0.99
    # ---- the whole CLI on the fake reader: files == those cells through the reference-shaped (golden-pinned) functions
    hooks.reset()
    fo.READS.clear()
    predict_wsi.main(args_f)
    assert len(fo.READS) == 6
    offs = np.concatenate([[0], np.cumsum(cells_f["n_pts"])])
    feats = []
    for i, c in enumerate(cells_f):
        centroid = np.round([c["cx"], c["cy"]], 2).tolist()
        feats.append(geojson.to_geojson_polygon(geojson.cell_dict(xy_f[offs[i]:offs[i + 1]].tolist(), int(c["cls"]), labels,
                                                                  c["area"], c["perimeter"], centroid)))
    feats = [geojson.apply_bounds_offset_to_feature(f, *bounds) for f in geojson.deduplicate(feats)]
    cont = json.load(open(next(out_f.glob("*contours.geojson"))))["features"]
    cent = json.load(open(next(out_f.glob("*centroids.geojson"))))["features"]
    assert 100 < len(cont) == len(feats) == len(cent) < len(cells_f)
    strip = lambda fs: [{k: v for k, v in f.items() if k != "id"} for f in fs]
    assert strip(cont) == strip(json.loads(json.dumps(feats)))
    assert strip(cent) == strip(json.loads(json.dumps(geojson.polygons_to_centroids(feats))))
    if bounds != (0.0, 0.0):
        xs = np.concatenate([np.asarray(f["geometry"]["coordinates"][0])[:, 0] for f in cont])
        assert xs.min() < 0                                   # cells left of the bounds origin come out negative, as in the reference


@pytest.mark.parametrize("geometry", ["conic-256-32-tta", "fp16-512-64"])
def test_predict_wsi_two_ranks_share_one_gpu_gloo(cuda, tmp_path, monkeypatch, geometry):
    """The N > 1 leg of BASELINE configs[3] (conic, --tta) and configs[4] (fp16, 512-px tiles) on a ONE-GPU box:
    ``--device cuda:0,0`` spawns two fresh rank processes on the same GPU (the spawn / static ``k % 2`` shard / all-gather of
    cell tables, vertex pools and tile indices / rank-0 outputs path of ``--device cuda:0,1``), with the collective carried
    by gloo (CPX_DIST_BACKEND) because one device cannot host two RCCL ranks.  The files must equal the single-rank run
    cell for cell -- including the de-duplication's order-dependent choices, which is what the canonical tile order of the
    gathered table is for."""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    monkeypatch.setenv("CPX_DIST_BACKEND", "gloo")
    from classpose_amd.entrypoints import predict_wsi
    if geometry == "conic-256-32-tta":
        slide, extra = "synthetic://1180x956?mpp=0.5&seed=51", dict(tile_size=256, overlap=32, tta=True, precision="bf16", batch_size=8)
    else:
        slide, extra = "synthetic://2000x1500?mpp=0.5&seed=52", dict(tile_size=512, overlap=64, tta=False, precision="fp16", batch_size=8)
    o1, o2 = tmp_path / "one", tmp_path / "two"
    predict_wsi.main(_reference_integration_args(slide, o1, device="cuda:0", **extra))
    predict_wsi.main(_reference_integration_args(slide, o2, device="cuda:0,0", **extra))
    feats = []
    for o in (o1, o2):
        fs = json.load(open(next(o.glob("*_cell_contours.geojson"))))["features"]
        feats.append([(f["geometry"]["coordinates"], f["properties"]["classification"], f["properties"]["measurements"]) for f in fs])
    assert len(feats[0]) > 100
    first = next((i for i, (a, b) in enumerate(zip(feats[0], feats[1])) if a != b), None)
    assert len(feats[0]) == len(feats[1]) and first is None, (                     # same cells, same order, same polygons
        len(feats[0]), len(feats[1]), first, None if first is None else (feats[0][first][1:], feats[1][first][1:]))
    cent = [json.load(open(next(o.glob("*_cell_centroids.geojson"))))["features"] for o in (o1, o2)]
    assert [f["geometry"] for f in cent[0]] == [f["geometry"] for f in cent[1]]


def test_predict_wsi_eight_ranks_share_one_gpu_gloo(cuda, tmp_path, monkeypatch):
    """BASELINE configs[3]'s rank count on a one-GPU box: ``--device cuda:0,0,0,0,0,0,0,0`` spawns EIGHT rank processes (conic,
    256 / 32, --tta; 20 tiles -> 2-3 per rank), the cell rows and tile indices are all-gathered, the vertex pools go to rank 0
    only (parallel.gather_to_root), and the two files equal the single-rank run feature for feature, in order."""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "1")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    monkeypatch.setenv("CPX_DIST_BACKEND", "gloo")
    from classpose_amd.entrypoints import predict_wsi
    slide, extra = "synthetic://1180x956?mpp=0.5&seed=53", dict(tile_size=256, overlap=32, tta=True, precision="bf16", batch_size=8)
    o1, o8 = tmp_path / "one", tmp_path / "eight"
    predict_wsi.main(_reference_integration_args(slide, o1, device="cuda:0", **extra))
    predict_wsi.main(_reference_integration_args(slide, o8, device="cuda:" + ",".join(["0"] * 8), **extra))
    feats = []
    for o in (o1, o8):
        fs = json.load(open(next(o.glob("*_cell_contours.geojson"))))["features"]
        feats.append([(f["geometry"]["coordinates"], f["properties"]["classification"], f["properties"]["measurements"]) for f in fs])
    assert len(feats[0]) > 100
    first = next((i for i, (a, b) in enumerate(zip(feats[0], feats[1])) if a != b), None)
    assert len(feats[0]) == len(feats[1]) and first is None, (                     # same cells, same order, same polygons
        len(feats[0]), len(feats[1]), first, None if first is None else (feats[0][first][1:], feats[1][first][1:]))
    cent = [json.load(open(next(o.glob("*_cell_centroids.geojson"))))["features"] for o in (o1, o8)]
    assert [f["geometry"] for f in cent[0]] == [f["geometry"] for f in cent[1]]
    # byte identity of the files apart from the uuid4 feature ids (the reference draws them at random too)
    import re
    strip = lambda b: re.sub(rb'"id": "[0-9a-f-]{36}"', b'"id": ""', b)
    for pat in ("*_cell_contours.geojson", "*_cell_centroids.geojson"):
        assert strip(next(o1.glob(pat)).read_bytes()) == strip(next(o8.glob(pat)).read_bytes())


def test_predict_wsi_cli_full_depth_network(cuda, tmp_path, monkeypatch):
    """The command line with the FULL 24-block ViT-L (random-init weights of the reference layout, as bench.py runs it) in
    flow-injection mode, BASELINE configs[1] geometry on a slide of 4 x 3 tiles: every nucleus inside the covered area comes
    out exactly once with the class of its id hash -- the property the depth-1 CLI tests check, at the depth that is benched."""
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_DEPTH", "24")
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    monkeypatch.setenv("CLASSPOSE_MODEL_DIR", str(tmp_path / "nomodels"))
    from classpose_amd.entrypoints import predict_wsi
    W, Hs, seed = 960, 730, 61
    args = predict_wsi.build_parser().parse_args([
        "--model_config", "conic", "--slide_path", f"synthetic://{W}x{Hs}?mpp=0.5&seed={seed}",
        "--output_folder", str(tmp_path), "--tile_size", "256", "--overlap", "32", "--batch_size", "32", "--device", "cuda:0"])
    assert args.precision == "bf16"
    predict_wsi.main(args)
    cont = json.load(open(next(tmp_path.glob("*contours.geojson"))))
    nx, ny = (W - 256) // 224 + 1, (Hs - 256) // 224 + 1
    assert (nx, ny) == (4, 3)
    cov_w, cov_h = (nx - 1) * 224 + 256, (ny - 1) * 224 + 256
    cx, cy, r, ident = synth.nuclei_in_region(seed, 0, 0, cov_w, cov_h)
    inner = (cx - r > 12) & (cx + r < cov_w - 12) & (cy - r > 12) & (cy + r < cov_h - 12)
    found = np.array([[m["value"] for m in f["properties"]["measurements"] if m["name"].startswith("centroid")]
                      for f in cont["features"]])
    from scipy.spatial import cKDTree
    d, idx = cKDTree(found).query(np.stack([cx[inner], cy[inner]], 1))
    assert np.all(d < 1.5) and len(np.unique(idx)) == inner.sum()
    names = ["Neutrophil", "Epithelial", "Lymphocyte", "Plasma cell", "Eosinophil", "Connective"]
    assert [cont["features"][j]["properties"]["classification"]["name"] for j in idx] == [names[int(i % np.uint64(6))] for i in ident[inner]]


def test_predict_wsi_command_line_forks_its_workers_and_equals_one_rank(cuda, tmp_path, monkeypatch):
    """The real command line (a fresh interpreter: ``python -m classpose_amd.entrypoints.predict_wsi ... --device cuda:0,0,0``): the parent has no GPU
    context and one thread, so it imports the tile loop once and FORKS its three workers (round 6; inside this test process the in-process calls above
    spawn instead, because the process holds a GPU context).  Its log must say so, and its files must equal the in-process single-rank run feature for
    feature -- the same tiles through the same kernels whichever way the ranks were started."""
    import subprocess
    env = dict(os.environ, CLASSPOSE_SYNTHETIC_WEIGHTS="1", CLASSPOSE_SYNTHETIC_DEPTH="1", CLASSPOSE_AMD_PLUGINS="classpose_amd.synth:flow",
               CLASSPOSE_MODEL_DIR=str(tmp_path / "nomodels"), CPX_DIST_BACKEND="gloo")
    env.pop("CLASSPOSE_START_METHOD", None)
    for k, v in env.items():
        if k.startswith(("CLASSPOSE_", "CPX_")):
            monkeypatch.setenv(k, v)
    from classpose_amd.entrypoints import predict_wsi
    slide = "synthetic://1180x956?mpp=0.5&seed=57"
    o1, o3 = tmp_path / "one", tmp_path / "three"
    o3.mkdir()
    predict_wsi.main(_reference_integration_args(slide, o1, device="cuda:0", tile_size=256, overlap=32, tta=False, precision="bf16", batch_size=8))
    r = subprocess.run([sys.executable, "-m", "classpose_amd.entrypoints.predict_wsi", "--model_config", "conic", "--slide_path", slide,
                        "--output_folder", str(o3), "--tile_size", "256", "--overlap", "32", "--precision", "bf16", "--device", "cuda:0,0,0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "once in the parent the ranks were forked from" in r.stderr, r.stderr[-1500:]
    feats = []
    for o in (o1, o3):
        fs = json.load(open(next(o.glob("*_cell_contours.geojson"))))["features"]
        feats.append([(f["geometry"]["coordinates"], f["properties"]["classification"], f["properties"]["measurements"]) for f in fs])
    assert len(feats[0]) > 100 and feats[0] == feats[1]
