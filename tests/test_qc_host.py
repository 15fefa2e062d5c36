"""Host side of GrandQC (a20): architecture tables, checkpoint loading, contours, patch assembly."""
import sys
import types

import numpy as np
import pytest
import torch

from classpose_amd import grandqc, qc_arch, roi, synth
from oracle import grandqc as og


def test_arch_tables_agree():
    """generated tables (product) == the oracle's literal copy; channel bookkeeping closes"""
    assert [dict(b) for b in qc_arch.encoder_blocks()] == og.A.encoder_blocks()
    assert qc_arch.decoder_schedule() == og.A.decoder_schedule()
    dec = qc_arch.decoder_blocks()
    out_ch = {f"f{k}": c for k, c in enumerate((320, 112, 40, 24, 32))}
    for name, xsrc, skips in qc_arch.decoder_schedule():
        ci, cs, co = dec[name]
        assert out_ch[xsrc] == ci and sum(out_ch[s] for s in skips) == cs
        out_ch[name] = co
    assert len(qc_arch.expected_shapes(2)) == 416
    sd = synth.make_grandqc_state_dict(8, 0)
    assert qc_arch.check_state_dict(sd) == 8
    bad = dict(sd); bad["decoder.blocks.x_0_1.conv1.0.weight"] = torch.zeros(128, 300, 3, 3)
    with pytest.raises(ValueError):
        qc_arch.check_state_dict(bad)


def test_pickled_module_checkpoint_without_its_package(tmp_path):
    """the artefact checkpoint is a pickled smp module: its tensors load without importing (or
    executing) the defining package; state-dict checkpoints load as they are"""
    pkg = types.ModuleType("fake_smp"); sub = types.ModuleType("fake_smp.unetplusplus")
    sys.modules["fake_smp"] = pkg; sys.modules["fake_smp.unetplusplus"] = sub

    class Head(torch.nn.Sequential):
        pass

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.Module()
            self.encoder.conv_stem = torch.nn.Conv2d(3, 32, 3, bias=False)
            self.encoder.bn1 = torch.nn.BatchNorm2d(32)
            self.segmentation_head = Head(torch.nn.Conv2d(16, 8, 3), torch.nn.Identity())
    Head.__module__ = Net.__module__ = "fake_smp.unetplusplus"
    Head.__qualname__, Net.__qualname__ = "Head", "Net"
    sub.Head, sub.Net = Head, Net
    model = Net()
    torch.save(model, tmp_path / "model.pth")
    ref = {k: v.clone() for k, v in model.state_dict().items()}
    del sys.modules["fake_smp"], sys.modules["fake_smp.unetplusplus"]
    sd = grandqc.load_qc_state_dict(str(tmp_path / "model.pth"), 8, 0)
    assert set(sd) == set(ref) and all(torch.equal(sd[k], ref[k]) for k in ref)
    assert "fake_smp" not in sys.modules
    torch.save(ref, tmp_path / "sd.pth")
    sd2 = grandqc.load_qc_state_dict(str(tmp_path / "sd.pth"), 8, 0)
    assert set(sd2) == set(ref)
    with pytest.raises(FileNotFoundError):
        grandqc.load_qc_state_dict(str(tmp_path / "missing.pth"), 2, 0)


def test_pickle_loader_refuses_code_execution(tmp_path):
    import pickle

    class Evil:
        def __reduce__(self):
            return (eval, ("1+1",))
    with open(tmp_path / "evil.pth", "wb") as f:
        pickle.dump(Evil(), f)
    with pytest.raises(Exception):
        grandqc.load_qc_state_dict(str(tmp_path / "evil.pth"), 2, 0)


def _ring_fill(cnts, parent, shape):
    """pixels on or inside an outer ring and not strictly inside one of its holes"""
    out = np.zeros(shape, bool)
    for i, c in enumerate(cnts):
        if parent[i] != -1:
            continue
        ring = np.concatenate([c, c[:1]]).astype(float)
        holes = [np.concatenate([cnts[j], cnts[j][:1]]).astype(float) for j in np.nonzero(parent == i)[0]]
        x0, y0 = c.min(0); x1, y1 = c.max(0)
        for y in range(y0, y1 + 1):
            for x in range(x0, x1 + 1):
                if len(c) == 1:
                    inside = (x, y) == tuple(c[0])
                else:
                    inside = roi._locate_in_ring(ring, x, y) >= 0
                if inside and not any(len(h) > 3 and roi._locate_in_ring(h, x, y) > 0 for h in holes):
                    out[y, x] = True
    return out


def test_find_contours_ccomp_known_shapes():
    m = np.zeros((7, 8), np.uint8); m[1:6, 1:6] = 1; m[3, 3] = 0; m[0, 7] = 1
    cnts, parent = grandqc.find_contours_ccomp(m)
    assert len(cnts) == 3 and list(parent).count(-1) == 2
    # filled rectangle: OpenCV's outer border starts at the top-left pixel and runs down first
    rect = [c for c, p in zip(cnts, parent) if p == -1 and len(c) == 4][0]
    assert rect.tolist() == [[1, 1], [1, 5], [5, 5], [5, 1]]
    hole = [c for c, p in zip(cnts, parent) if p != -1][0]
    assert hole[0].tolist() == [2, 3] and sorted(hole.tolist()) == sorted([[2, 3], [3, 4], [4, 3], [3, 2]])
    single = [c for c, p in zip(cnts, parent) if p == -1 and len(c) == 1][0]
    assert single.tolist() == [[7, 0]]
    assert grandqc.contour_area(rect) == 16.0
    assert grandqc.find_contours_ccomp(np.zeros((5, 5), np.uint8))[0] == []


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_find_contours_ccomp_reconstructs_mask(seed):
    """outer rings minus hole rings rasterise back to the mask; counts match component / hole counts"""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    m = ndimage.gaussian_filter(rng.standard_normal((48, 64)), 2.0 + seed) > 0.02
    m = m.astype(np.uint8)
    cnts, parent = grandqc.find_contours_ccomp(m)
    n_comp = ndimage.label(m, structure=np.ones((3, 3)))[1]
    bg, n_bg = ndimage.label(m == 0)                                     # 4-connected background
    border = set(np.unique(np.concatenate([bg[0], bg[-1], bg[:, 0], bg[:, -1]]))) - {0}
    assert (parent == -1).sum() == n_comp and (parent != -1).sum() == n_bg - len(border)
    assert np.array_equal(_ring_fill(cnts, parent, m.shape), m.astype(bool))
    assert all(parent[p] == -1 for p in parent if p != -1)                # two-level hierarchy


def test_tissue_assembly_matches_reference_loop():
    """our indexed assembly == the reference's np.concatenate loop for every overhang case"""
    rng = np.random.default_rng(0)

    class FakeNet:
        device = "cpu"

    def predict(patch):                                                   # deterministic stand-in "network"
        return (patch[..., 0] > patch[..., 1]).astype(np.int8)
    orig = grandqc._run_patches
    grandqc._run_patches = lambda net, patches: [predict(p) for p in patches]
    try:
        for (h, w) in [(100, 130), (64, 64), (70, 128), (128, 128), (129, 200), (40, 30), (64, 150)]:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            got = grandqc.tissue_class_map(img, FakeNet(), 64)
            want = og.tissue_class_map_ref(img, predict, 64)
            assert got.shape == (h, w) and np.array_equal(got, want), (h, w)
        img = rng.integers(0, 256, (200, 150, 3), dtype=np.uint8)
        tm = (rng.random((200, 150)) > 0.4).astype(np.uint8); tm[:64, :64] = 0
        got = grandqc.artefact_class_map(img, tm, FakeNet(), 64)
        assert np.array_equal(got, og.artefact_mask_ref(img, tm, predict, 64))
    finally:
        grandqc._run_patches = orig


def test_tissue_contours_area_filter_and_holes():
    cm = np.ones((40, 60), np.int8)                     # class 1 = background, 0 = tissue
    cm[5:25, 5:30] = 0; cm[10:15, 10:15] = 1            # 20x25 tissue block with a 5x5 hole
    cm[30:33, 50:53] = 0                                # 9-px speck
    filtered, cnts = grandqc.tissue_contours(cm, 10, min_area=1000, scaling=(8.0, 4.0))
    assert filtered.sum() == 20 * 25 - 25 and len(cnts) == 1
    c = next(iter(cnts.values()))
    assert c["contour"][0].tolist() == [5 * 8.0, 5 * 4.0] and np.array_equal(c["contour"][0], c["contour"][-1])
    assert len(c["holes"]) == 1 and c["holes"][0].min(0).tolist() == [9 * 8.0, 9 * 4.0]
    _, cnts0 = grandqc.tissue_contours(cm, 10, min_area=0, scaling=(1.0, 1.0))
    assert len(cnts0) == 2
    gj = grandqc._cnts_to_geojson(cnts, "tissue", [0, 0, 0])
    assert gj["features"][0]["geometry"]["type"] == "Polygon" and len(gj["features"][0]["geometry"]["coordinates"]) == 2


def test_resize_nearest_and_jpeg():
    m = np.arange(12, dtype=np.uint8).reshape(3, 4)
    up = grandqc.resize_nearest(m, 8, 6)
    assert up.shape == (6, 8) and np.array_equal(up[::2, ::2], m) and np.array_equal(up[1::2, 1::2], m)
    assert np.array_equal(grandqc.resize_nearest(m, 2, 3), m[:, ::2])
    img = synth.render_region(5, 0, 0, 96, 64)
    j = grandqc.simulate_jpeg_compression(img)
    assert j.shape == img.shape and j.dtype == np.uint8 and not np.array_equal(j, img)
    assert np.abs(j.astype(int) - img.astype(int)).mean() < 12


def test_densities_match_reference_golden(golden):
    """calculate_cellular_densities / get_artefact_class_id vs vectors minted from the reference itself
    (tests/golden/make_golden_outputs.py)"""
    import json, os
    from classpose_amd import outputs
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_outputs.json")))
    for case in g["global_cases"]:
        df = outputs.calculate_cellular_densities(case["cells"], case["tissue"], case["artefact"], case["mpp_x"],
                                                  case["mpp_y"], g["labels"])
        assert df.to_dict("records") == case["rows"]
    rc = g["roi_case"]
    df = outputs.calculate_cellular_densities(rc["cells"], rc["tissue"], rc["artefact"], rc["mpp_x"], rc["mpp_y"], g["labels"])
    assert df.to_dict("records") == rc["rows"] and list(df.columns) == ["region", "cell_class", "count", "density"]
    assert {n: roi.get_artefact_class_id(n) for n in g["artefact_class_ids"]} == g["artefact_class_ids"]


def test_map_cells_to_roi_classes_priority():
    from classpose_amd import outputs

    def cell(x, y, name="a"):
        ring = [[x - 1, y - 1], [x + 1, y - 1], [x + 1, y + 1], [x - 1, y + 1], [x - 1, y - 1]]
        return {"geometry": {"coordinates": [ring]}, "properties": {"classification": {"name": name}}}
    sq = lambda x0, y0, s: roi.RoiPolygon([[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s]])
    classes = {"Tumour": [sq(0, 0, 100)], "Stroma": [sq(50, 50, 100)], "Necrosis": []}
    cells = [cell(10, 10), cell(75, 75), cell(140, 140), cell(300, 300)]
    m = outputs.map_cells_to_roi_classes(cells, classes)
    assert [len(m[k]) for k in ("Tumour", "Stroma", "Necrosis")] == [2, 1, 0]
    m = outputs.map_cells_to_roi_classes(cells, classes, ["Stroma", "Missing"])
    assert [len(m[k]) for k in ("Tumour", "Stroma", "Necrosis")] == [1, 2, 0]


def test_pickle_loader_allow_list_blocks_getattr_chain(tmp_path):
    """the ADVICE exploit shape: GLOBAL torch.serialization.os -> builtins.getattr -> REDUCE.  Neither a
    whole-module torch prefix nor getattr is reachable: the first global is an inert stub, the second is refused"""
    import pickle
    import pickletools  # noqa: F401
    payload = (b"\x80\x02cbuiltins\ngetattr\n(ctorch.serialization\nos\nX\x06\x00\x00\x00systemtR(X\x0b\x00\x00\x00echo pwned!tR.")
    p = tmp_path / "chain.pth"
    p.write_bytes(payload)
    with pytest.raises(Exception) as ei:
        grandqc.load_qc_state_dict(str(p), 2, 0)
    assert "refusing builtins.getattr" in str(ei.value) or isinstance(ei.value, pickle.UnpicklingError)
    up = grandqc._StateOnlyUnpickler(__import__("io").BytesIO(b""))
    stub = up.find_class("torch.serialization", "os")
    assert issubclass(stub, grandqc._Stub) and stub.__module__ == "torch.serialization"
    assert issubclass(up.find_class("functools", "partial"), grandqc._Stub)
    assert issubclass(up.find_class("torch.nn.modules.conv", "Conv2d"), grandqc._Stub)
    with pytest.raises(pickle.UnpicklingError):
        up.find_class("os", "system")


def test_tissue_pipeline_equals_reference_detect_tissue_wsi():
    """a20 host orchestration: ``detect_tissue_wsi`` (grandqc/wsi_tissue_detection.py:32-329) run by the reference itself
    on fabricated slides with an elementwise stand-in network (tests/golden/make_golden_qc.py) against the product's
    ``tissue_class_map`` -> ``tissue_contours`` -> ``_cnts_to_geojson`` (-> ``_shift_outputs``) with the same stand-in:
    the stitched class map's tissue mask, every contour and hole in level-0 coordinates, the GeoJSON features."""
    import json
    import os
    import sys
    import types

    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_qc as mq
    from classpose_amd import grandqc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_qc.npz"))

    class FakeNet:                                      # QcNet's interface: uint8 (n, 512, 512, 3) -> int8 class maps
        device = torch.device("cpu")

        def forward(self, patches):
            out = []
            for p in patches.numpy():
                x = mq.preprocessing_fn(p).transpose(2, 0, 1).astype("float32")          # get_preprocessing + to_tensor_x
                out.append(np.argmax(mq.fake_logits(x), axis=0).astype("int8"))
            return torch.from_numpy(np.stack(out))

    for k in range(int(g["td_n"])):
        w0, h0, min_area, shift, seed, tw, th = (int(v) for v in g[f"td_{k}_cfg"])
        props = json.loads(str(g[f"td_{k}_props"]))
        slide = types.SimpleNamespace(properties=props, level_dimensions=[(w0, h0)])
        assert grandqc.extract_slide_info(slide, 10)[3] == (tw, th)
        image = mq.thumbnail(seed, tw, th)
        class_map = grandqc.tissue_class_map(image, FakeNet(), 512)
        filtered, cnts = grandqc.tissue_contours(class_map, 10, min_area, (w0 / tw, h0 / th))
        want_mask = np.unpackbits(g[f"td_{k}_mask"])[: th * tw].reshape(th, tw)
        assert np.array_equal(filtered, want_mask)
        gj = grandqc._cnts_to_geojson(cnts, "tissue", [0, 0, 0])
        if shift:
            grandqc._shift_outputs(cnts, gj, float(props["openslide.bounds-x"]), float(props["openslide.bounds-y"]))
        assert sorted(cnts.keys()) == g[f"td_{k}_keys"].tolist()
        for i, key in enumerate(sorted(cnts.keys())):
            assert np.array_equal(cnts[key]["contour"], g[f"td_{k}_c{i}"])
            assert len(cnts[key]["holes"]) == int(g[f"td_{k}_c{i}_nh"])
            for j, hct in enumerate(cnts[key]["holes"]):
                assert np.array_equal(hct, g[f"td_{k}_c{i}_h{j}"])
        for f in gj["features"]:
            f["id"] = "uuid"
        assert json.loads(json.dumps(gj)) == json.loads(str(g[f"td_{k}_geojson"]))


def test_artefact_pipeline_equals_reference_detect_artefacts_wsi():
    """a20 host orchestration, artefact half: ``detect_artefacts_wsi`` (grandqc/wsi_artefact_detection.py:56-348) run by
    the reference itself (tissue step stood in by a given mask, 7-class elementwise stand-in network) against the product's
    ``resize_nearest`` -> ``artefact_class_map`` -> ``artefact_contours`` (-> ``_shift_outputs``) and the LANCZOS overview
    map: the tissue-gated, padded class mask, the filter polygons with their holes, every GeoJSON feature, the class
    names and colours."""
    import json
    import os
    import sys

    import torch
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_qc as mq
    from classpose_amd import grandqc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_qc.npz"))
    assert json.loads(str(g["ad_colors"])) == [list(c) for c in grandqc.ARTIFACT_COLORS]
    assert {int(k): v for k, v in json.loads(str(g["ad_names"])).items()} == dict(grandqc.ARTIFACT_CLASS_MAPPING)

    class FakeNet7:
        device = torch.device("cpu")

        def forward(self, patches):
            out = []
            for p in patches.numpy():
                x = mq.preprocessing_fn(p).transpose(2, 0, 1).astype("float32")
                out.append(np.argmax(mq.fake_logits7(x), axis=0).astype("int8"))
            return torch.from_numpy(np.stack(out))

    for k in range(int(g["ad_n"])):
        w0, h0, shift, seed, tw, th, mw, mh = (int(v) for v in g[f"ad_{k}_cfg"])
        props = json.loads(str(g[f"ad_{k}_props"]))
        image = mq.artefact_thumbnail(seed, tw, th)
        tmask = mq.tissue_mask_10(seed, mw, mh)
        tissue_art = grandqc.resize_nearest(tmask, tw, th)
        amask = grandqc.artefact_class_map(image, tissue_art, FakeNet7(), 512)
        assert np.array_equal(amask.astype(np.uint8), g[f"ad_{k}_mask"])
        cnts, gj = grandqc.artefact_contours(amask, (w0 / tw, h0 / th))
        if shift:
            grandqc._shift_outputs(cnts, gj, float(props["openslide.bounds-x"]), float(props["openslide.bounds-y"]))
        keys = json.loads(str(g[f"ad_{k}_keys"]))
        assert sorted(cnts.keys()) == keys
        for i, key in enumerate(keys):
            assert np.array_equal(cnts[key]["contour"], g[f"ad_{k}_c{i}"])
            assert len(cnts[key]["holes"]) == int(g[f"ad_{k}_c{i}_nh"])
            for j, hct in enumerate(cnts[key]["holes"]):
                assert np.array_equal(hct, g[f"ad_{k}_c{i}_h{j}"])
        for f in gj["features"]:
            f["id"] = "uuid"
        assert json.loads(json.dumps(gj)) == json.loads(str(g[f"ad_{k}_geojson"]))
        amap = Image.fromarray(grandqc.make_class_map(amask, grandqc.ARTIFACT_COLORS)).resize(
            (int(tw * 50 / 512), int(th * 50 / 512)), Image.Resampling.LANCZOS)
        assert np.array_equal(np.array(amap), g[f"ad_{k}_map"])
