"""CPU: the oracle against the golden vectors minted from the reference itself
(tests/golden/make_golden.py) and against the libraries the reference bottoms out in."""
import numpy as np
import pytest
import torch

from oracle import classmask, cref, dynamics, net, tiling
from classpose_amd import synth


def test_remove_border_instances_golden(golden):
    npz, _ = golden
    for i in range(int(npz["rbi_n"])):
        out = classmask.remove_border_instances(npz[f"rbi_in_{i}"].copy())
        assert np.array_equal(out, npz[f"rbi_out_{i}"]), i


def test_remove_border_instances_reference_known_answers():
    # the 9 cases of /root/reference/tests/test_remove_border_instances.py, restated as data
    m = np.zeros((6, 6), np.int64)
    m[0:3, 0:3] = 1; m[0:3, 3:6] = 2; m[2:4, 2:4] = 3; m[3:6, 3:6] = 4
    r = classmask.remove_border_instances(m.copy())
    assert np.all(r[r != 0] == 3)
    assert (r[2, 2], r[2, 3], r[3, 2], r[3, 3]) == (3, 0, 0, 0) or (r[2, 2], r[3, 3]) == (3, 0)
    a = np.zeros((4, 4), np.int64); a[0:2] = 1; a[2:4] = 2
    assert np.all(classmask.remove_border_instances(a) == 0)
    assert np.all(classmask.remove_border_instances(np.zeros((5, 5), np.int64)) == 0)
    b = np.zeros((5, 5), np.int64); b[1:4, 1:4] = 7
    rb = classmask.remove_border_instances(b)
    assert np.all(rb[1:4, 1:4] == 7) and rb[0].sum() == 0 and rb[:, 0].sum() == 0
    cls = np.zeros_like(m)
    for k, v in ((1, 1), (2, 2), (3, 3), (4, 1)):
        cls[m == k] = v
    c = classmask.remove_border_instances(np.stack([m, cls], -1))
    assert set(np.unique(c[..., 0])) == {0, 3} and set(np.unique(c[..., 1])) == {0, 3}


def test_compute_class_masks_golden(golden):
    npz, _ = golden
    for i in range(int(npz["ccm_n"])):
        cm, uniq = classmask.compute_class_masks(npz[f"ccm_masks_{i}"], npz[f"ccm_logits_{i}"])
        assert np.array_equal(cm, npz[f"ccm_out_{i}"]), i
        assert np.array_equal(uniq, npz[f"ccm_uniq_{i}"]), i
        assert cm.dtype == npz[f"ccm_out_{i}"].dtype


def test_unaugment_class_tiles_golden(golden):
    npz, _ = golden
    out = tiling.unaugment_class_tiles(npz["uct_in"].copy())
    assert np.array_equal(out, npz["uct_out"])


def test_unet_golden(golden):
    npz, _ = golden
    sd = {"h." + k[len("unet_sd__"):]: torch.from_numpy(npz[k]) for k in npz.files
          if k.startswith("unet_sd__")}
    y = net.unet_forward(sd, "h.", torch.from_numpy(npz["unet_x"]), 2).numpy()
    assert np.allclose(y, npz["unet_y"], rtol=0, atol=1e-6)
    # four encoder levels, ragged widths [6, 10, 12, 20] (the reference's UNet run by make_golden.py)
    sd = {"h." + k[len("unet4_sd__"):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith("unet4_sd__")}
    y = net.unet_forward(sd, "h.", torch.from_numpy(npz["unet4_x"]), 4).numpy()
    assert y.shape == npz["unet4_y"].shape == (1, 24, 32, 32)
    assert np.allclose(y, npz["unet4_y"], rtol=0, atol=1e-6)


def test_get_coords_golden(golden):
    _, js = golden
    for key, exp in js["get_coords"].items():
        if key.startswith("rect"):
            got = tiling.get_coords(256, 32, (1000, 700), 2.0)
            assert [[list(c), t] for c, t in got] == exp["all"]
            continue
        dim, tile, ov = map(int, key.split("_"))
        got = tiling.get_coords(tile, ov, (dim, dim), 1.0)
        assert len(got) == exp["n"]
        assert [[list(c), t] for c, t in got[:3]] == exp["first"]
        assert [[list(c), t] for c, t in got[-3:]] == exp["last"]
    assert len(tiling.get_coords(256, 32, (10000, 10000), 1.0)) == 1936
    assert len(tiling.get_coords(256, 32, (40000, 40000), 1.0)) == 31684


def test_deduplicate_golden(golden):
    _, js = golden
    pts = np.array(js["geojson"]["points"])
    kept = classmask.deduplicate_indices(pts[:, :2], pts[:, 2])
    assert kept == js["geojson"]["kept_ids"]


def test_c_follow_flows_equals_torch():
    """oracle/c/dynamics_ref.c == the literal torch grid_sample loop, bit for bit."""
    rng = np.random.default_rng(3)
    for H, W in ((64, 50), (96, 128)):
        dP = (rng.standard_normal((2, H, W)) * 3).astype(np.float32)
        cp = rng.standard_normal((H, W)).astype(np.float32)
        inds = np.nonzero(cp > 0)
        f = dP * (cp > 0) / 5.0
        a = dynamics.follow_flows(f, inds, 60).numpy()
        b = cref.follow_flows(f, inds, 60)
        assert np.array_equal(a, b)
    dP, cp, _, _ = synth.analytic_fields(1234, 300, 40, 128, 128, 7)
    inds = np.nonzero(cp > 0)
    f = dP * (cp > 0) / 5.0
    assert np.array_equal(dynamics.follow_flows(f, inds, 200).numpy(), cref.follow_flows(f, inds, 200))


def test_normalize_matches_numpy_percentile():
    img = synth.render_region(1234, 0, 0, 96, 80)
    x = tiling.normalize_img(img[None])
    for c in range(3):
        ch = img[..., c].astype(np.float32)
        x01, x99 = np.percentile(ch, 1), np.percentile(ch, 99)
        assert np.array_equal(x[0, ..., c], (ch - x01) / (x99 - x01))
    const = np.full((1, 16, 16, 3), 200, np.uint8)
    assert np.array_equal(tiling.normalize_img(const), const.astype(np.float32))   # ptp == 0 quirk


def test_tiling_roundtrip_identity():
    """average_tiles(make_tiles(x)) == x (to float32 rounding) with and without TTA."""
    rng = np.random.default_rng(0)
    x = rng.random((1, 256, 256, 3)).astype(np.float32)

    def fw(img):   # "network" that returns its input channels as flows and 2 class maps
        return img.copy(), img[:, :2].copy()
    for aug in (False, True):
        IMGa, geom = tiling.subtile_batch(x, 256, aug)
        assert IMGa.shape[0] == (9 if aug else 4)
        y, yc = fw(IMGa)
        if aug:   # a real net sees flipped tiles: emulate the sign convention of unaugment_tiles
            y = y.reshape(geom["ny"], geom["nx"], 3, 256, 256)
            for j in range(geom["ny"]):
                for i in range(geom["nx"]):
                    if j % 2 == 0 and i % 2 == 1:
                        y[j, i, 0] *= -1
                    elif j % 2 == 1 and i % 2 == 0:
                        y[j, i, 1] *= -1
                    elif j % 2 == 1 and i % 2 == 1:
                        y[j, i, :2] *= -1
            y = y.reshape(-1, 3, 256, 256)
        yf, ycf = tiling.blend_subtiles(y, yc, geom, aug)
        assert np.allclose(yf, x[0].transpose(2, 0, 1), atol=2e-6)
        assert np.allclose(ycf, x[0].transpose(2, 0, 1)[:2], atol=2e-6)


def test_compute_masks_analytic_discs():
    dP, cp, lg, nfull = synth.analytic_fields(1234, 512, 256, 192, 160, 7)
    m = dynamics.compute_masks(dP, cp)
    assert m.dtype == np.uint16 and m.shape == (160, 192)
    assert nfull <= m.max() <= nfull + 40          # + partial discs at the border
    assert set(np.unique(m)) == set(range(m.max() + 1))   # renumbered, no gaps
    cm, _ = classmask.compute_class_masks(m, lg)
    assert cm.max() <= 6 and cm[m == 0].max() == 0
    assert np.all(dynamics.compute_masks(dP, np.full_like(cp, -1.0)) == 0)   # no active pixel


def test_size_filter_positional_quirk_is_kept():
    """fill_holes_and_remove_small_masks indexes unique()[1][1:] by position (cellpose quirk)."""
    m = np.zeros((12, 40), np.uint16)
    m[1:6, 1:6] = 1        # 25 px
    m[1:6, 10:15] = 2      # 25 px
    m[1:6, 20:25] = 4      # 25 px   (label 3 missing, as after remove_bad_flow_masks)
    m[1:3, 30:33] = 5      # 6 px -> small, sits at position 3 -> label VALUE 4 is removed
    out = dynamics.fill_holes_and_remove_small_masks(m, 15)
    assert out[3, 22] == 0            # the innocent label 4 was deleted by the first filter
    assert out[2, 31] == 0            # label 5 is dropped by the second filter
    assert out.max() == 2


def test_net_oracle_shapes_and_dtype():
    sd = synth.make_state_dict(7, None, depth=1, seed=1)
    x = torch.from_numpy(np.random.default_rng(0).random((1, 3, 256, 256)).astype(np.float32))
    out = net.class_transformer_forward(sd, x)
    assert out.shape == (1, 10, 256, 256) and out.dtype == torch.float32
    sdb = {k: v.to(torch.bfloat16) if v.is_floating_point() else v for k, v in sd.items()}
    outb = net.class_transformer_forward(sdb, x, torch.bfloat16)
    rel = (outb - out).norm() / out.norm()
    assert rel < 3e-2


def test_host_logic_matches_reference_goldens():
    """a1 / a5 host logic against outputs of the reference's own functions (tests/golden/make_golden_host.py):
    get_slide_resolution (utils.py:285-331) incl. its error messages, infer_structure (predict_wsi.py:1377-1419) on
    the key layouts of synthetic checkpoints the reference itself loaded, resolve_precision (models.py:37-69)."""
    import json
    import os
    import types

    import torch

    from classpose_amd import engine, models, wsi
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    for c in g["get_slide_resolution"]:
        slide = types.SimpleNamespace(properties=c["properties"])
        if "error" in c:
            with pytest.raises(ValueError) as e:
                wsi.get_slide_resolution(slide)
            assert c["error"] == "ValueError" and str(e.value) == c["message"]
        else:
            assert list(wsi.get_slide_resolution(slide)) == c["mpp"]           # same float operations, same bits
    for c in g["infer_structure"]:
        sd = {k: torch.empty(shape, device="meta") for k, shape in c["keys"].items()}
        fts, ncls, depth = engine.NetWeights.infer_structure(sd)
        assert (fts, ncls, depth) == (c["fts"], c["n_classes"], 1)
    names = {"torch.float32": "fp32", "torch.float16": "fp16", "torch.bfloat16": "bf16"}
    for c in g["resolve_precision"]:
        if "error" in c:
            with pytest.raises(ValueError) as e:
                models.resolve_precision(c["precision"], torch.device("cpu"))
            assert str(e.value) == c["message"]
        else:
            assert models.resolve_precision(c["precision"], torch.device("cpu")) == names[c["dtype"]]


def test_grandqc_host_helpers_match_reference_goldens():
    """a20 host helpers against the reference's wsi_qc_helpers (extract_slide_info :26-46, create_geojson_feature
    :49-88, make_class_map :123-150), goldens from tests/golden/make_golden_host.py."""
    import json
    import os
    import types

    from classpose_amd import grandqc
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    for c in g["extract_slide_info"]:
        slide = types.SimpleNamespace(level_dimensions=[tuple(c["dims"])], properties=c["properties"])
        w, h, mpp, td = grandqc.extract_slide_info(slide, c["mpp_model"])
        assert [w, h, mpp, list(td)] == c["out"]
    for c in g["create_geojson_feature"]:
        f = grandqc.create_geojson_feature(np.array(c["contour"]), np.array(c["scaling"]), "Fold", [255, 99, 71])
        if c["feature"] is None:
            assert f is None
        else:
            f = dict(f, id="uuid")
            assert json.loads(json.dumps(f)) == c["feature"]
    c = g["make_class_map"]
    assert grandqc.make_class_map(np.array(c["mask"]), c["colors"]).tolist() == c["rgb"]


def test_boundary_helpers_match_reference_goldens():
    """(b) drop-in boundary: output file names (utils.py:29-93, incl. the error text) and --device parsing
    (utils.py:118-148) against outputs of the reference's own functions."""
    import json
    import os

    from classpose_amd.entrypoints import predict_wsi as pw
    from classpose_amd.grandqc import wsi_tissue_detection as td
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    for c in g["output_filename"]:
        if "error" in c:
            with pytest.raises(ValueError) as e:
                pw.get_geojson_output_filename(c["kind"], c["base"])
            assert str(e.value) == c["message"]
        else:
            assert pw.get_geojson_output_filename(c["kind"], c["base"]) == c["name"]
    for c in g["output_path_from_prefix"]:
        assert str(td.get_geojson_output_path_from_prefix(c["prefix"], c["kind"])) == c["path"]
    for c in g["get_device"]:
        assert [str(d) for d in pw.get_device(c["device"])] == c["devices"]


def test_model_configs_match_reference_goldens(tmp_path):
    """--model_config surface against the reference's own table and YAML loader (model_configs.py:20-148)."""
    import json
    import os

    from classpose_amd import model_configs as mc
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    ours = {k: dict(v, path=os.path.relpath(v["path"], str(mc.ROOT_MODEL_DIR))) for k, v in mc.DEFAULT_MODEL_CONFIGS.items()}
    assert ours == g["default_model_configs"]
    for c in g["model_config_yaml"]:
        yp = tmp_path / "c.yaml"
        yp.write_text(c["yaml"])
        assert mc.ModelConfig.load_from_yaml(str(yp)).model_dump() == c["config"]


def test_cli_arguments_match_reference_goldens():
    """(b) the console script's argument surface: for the same argv, our parser hands run the same namespace as the
    reference's main_with_args hands its main() (predict_wsi.py:1891-2021): names, defaults, types, list arguments,
    --tta / --no-tta style switches."""
    import json
    import os

    from classpose_amd.entrypoints import predict_wsi as pw
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    for c in g["cli"]:
        assert vars(pw.build_parser().parse_args(c["argv"])) == c["args"]
    from classpose_amd.entrypoints import predict_wsi_cpsam as pc            # predict_wsi_cpsam.py:677-812
    for c in g["cli_cpsam"]:
        assert vars(pc.build_parser().parse_args(c["argv"])) == c["args"]


def test_plan_slide_matches_reference_init_slide():
    """a1 / a2: ``SlideLoader._init_slide`` (predict_wsi.py:220-278) run by the reference itself on fabricated multi-level
    slides -- mpp, bounds, level choice, ``ts`` / ``resize_factor`` as its manager.Value("f") slots return them (full doubles: a real Manager is used), the
    read-tile geometry and the whole tile grid -- against ``wsi.plan_slide``."""
    import json
    import os

    from classpose_amd import wsi
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    for c in g["init_slide"]:
        class FakeSlide:
            properties = c["props"]
            level_dimensions = [tuple(d) for d in c["dims"]]
            level_downsamples = list(c["down"])

            def get_best_level_for_downsample(self, ds):
                best = 0
                for i, d in enumerate(self.level_downsamples):
                    if d <= ds:
                        best = i
                return best
        p = wsi.plan_slide(FakeSlide(), c["tile"], c["overlap"], c["train_mpp"])
        assert list(p.mpp) == c["mpp"]
        assert list(p.bounds) == c["bounds"]
        assert p.level == c["level"] and list(p.slide_dim) == c["slide_dim"]
        assert p.ts == c["ts"] and p.resize_factor == c["resize_factor"]                  # exactly
        # main() reads mpp_x / mpp_y back from the loader's shared slots and derives the polygon scale from THOSE
        # (predict_wsi.py:1517-1524); the golden holds the slot values (doubles)
        assert list(p.mpp) == [c["mpp_x"], c["mpp_y"]]
        assert p.polygon_scale == min(c["train_mpp"] / c["mpp_x"], c["train_mpp"] / c["mpp_y"])
        assert len(p.coords) == c["n_coords"]
        as_list = lambda cs: [[list(map(int, x[0])), int(x[1])] for x in cs]
        assert as_list(p.coords[:3]) == c["coords_head"] and as_list(p.coords[-2:]) == c["coords_tail"]
        assert [sum(x[0][0] for x in p.coords), sum(x[0][1] for x in p.coords)] == c["coords_sum"]


def test_resized_shape_matches_reference():
    """a3: the output size of ``resize_tile_to_target_mpp`` (predict_wsi.py:102-123, Python's round-half-even) as observed
    from the reference through a recording stand-in for cv2.resize."""
    import json
    import os

    from classpose_amd import ops
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")) as f:
        g = json.load(f)
    for c in g["resize_shape"]:
        assert list(ops.resized_shape(c["h"], c["w"], c["factor"])) == c["out_hw"]
        assert c["called_resize"] == (c["factor"] != 1.0)
