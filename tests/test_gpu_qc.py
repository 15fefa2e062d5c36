"""GPU parity: GrandQC UNet++/EfficientNet-B0 forward (cpx_qc_forward) vs the torch-CPU oracle."""
import numpy as np
import pytest
import torch

from classpose_amd import grandqc, synth
from oracle import grandqc as og

pytestmark = pytest.mark.gpu


def _oracle_logits(sd, patches):
    with torch.no_grad():
        x = torch.cat([og.preprocess(p) for p in patches])
        return og.forward(sd, x).permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("n_classes,nB,H,W,seed", [(2, 1, 512, 512, 1), (8, 2, 256, 320, 2), (2, 3, 64, 96, 3)])
def test_qc_forward_matches_oracle(cuda, n_classes, nB, H, W, seed):
    """float32 logits: rel-L2 <= 2e-5 of the oracle (BatchNorm folding + summation order are the only
    differences); class maps identical wherever the oracle's top-2 margin exceeds 1e-3"""
    sd = synth.make_grandqc_state_dict(n_classes, seed)
    patches = np.stack([synth.render_region(40 + seed, 700 * i, 33 * i, W, H) for i in range(nB)])
    patches[0, : H // 2, : W // 3] = 245                      # a flat background region
    net = grandqc.QcNet.from_state_dict(sd, cuda)
    cls, logits = net.forward(torch.from_numpy(patches).to(cuda), return_logits=True)
    ref = _oracle_logits(sd, patches)
    got = logits.cpu().numpy()
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 2e-5, err
    assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    ref_cls = ref.argmax(-1)
    srt = np.sort(ref, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert decided.mean() > 0.99
    assert np.array_equal(cls.cpu().numpy()[decided], ref_cls[decided].astype(np.int8))
    # np.argmax tie rule on the device's own logits: bit-exact
    assert np.array_equal(cls.cpu().numpy(), got.argmax(-1).astype(np.int8))
    # deterministic
    cls2 = net.forward(torch.from_numpy(patches).to(cuda))
    assert torch.equal(cls, cls2)


def test_detect_tissue_and_artefacts_api_and_cli(cuda, tmp_path, monkeypatch):
    """what the reference's tests/test_grandqc_integration.py asserts (types, FeatureCollection), on a
    synthetic slide with synthetic weights, plus the stand-alone CLIs' output files"""
    import json
    from classpose_amd import wsi
    from classpose_amd.grandqc import wsi_artefact_detection, wsi_tissue_detection
    monkeypatch.setenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "1")
    uri = "synthetic://6000x5200?mpp=0.5&seed=8"
    slide = wsi.WSIReader(uri)
    image, mask, filled, cnts, geojson, mpp_td = grandqc.detect_tissue_wsi(slide, str(tmp_path / "td.pth"), device=cuda)
    assert isinstance(image, np.ndarray) and isinstance(mask, np.ndarray) and isinstance(filled, np.ndarray)
    assert isinstance(cnts, dict) and geojson["type"] == "FeatureCollection" and mpp_td == 10
    assert image.shape[:2] == mask.shape == filled.shape == (260, 300)
    # against the oracle run through the reference's literal patch loop (same JPEG'd thumbnail)
    sd = synth.make_grandqc_state_dict(2, 101)
    want = og.tissue_class_map_ref(image, lambda p: og.predict_mask(sd, p))
    net = grandqc.QcNet.from_state_dict(sd, cuda)
    got = grandqc.tissue_class_map(image, net)
    assert got.shape == want.shape and (got != want).mean() < 2e-3
    amask, amap, acnts, agj = grandqc.detect_artefacts_wsi(slide, str(tmp_path / "art.pth"), device=cuda,
                                                           model_td_path=str(tmp_path / "td.pth"))
    assert isinstance(amask, np.ndarray) and isinstance(amap, np.ndarray) and isinstance(acnts, dict)
    assert agj["type"] == "FeatureCollection" and amask.shape == (2600, 3000) and amap.shape[2] == 3
    assert set(np.unique(amask)) <= set(range(8))
    wsi_tissue_detection.main(["--slide_path", uri, "--output_path", str(tmp_path / "t"), "--model_path",
                               str(tmp_path / "td.pth"), "--device", "cuda:0"])
    for suffix in ("_image.png", "_mask.png", "_filled_class_map.png", "_tissue_contours.geojson"):
        assert (tmp_path / ("t" + suffix)).exists(), suffix
    assert json.load(open(tmp_path / "t_tissue_contours.geojson"))["type"] == "FeatureCollection"
    wsi_artefact_detection.main(["--slide_path", uri, "--output_path", str(tmp_path / "a"), "--model_art_path",
                                 str(tmp_path / "art.pth"), "--model_td_path", str(tmp_path / "td.pth"),
                                 "--device", "cuda:0"])
    for suffix in ("_artefact_map.png", "_artefact_mask.png", "_artefact_contours.geojson"):
        assert (tmp_path / ("a" + suffix)).exists(), suffix
