"""CPU: the oracle's network glue against vectors minted by the REFERENCE'S OWN functions
(tests/golden/make_golden_network.py imports /root/reference/src under the stub finder and calls
classpose.vit_sam.flash_forward, ClassTransformer.forward and classpose.core.run_net)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden_network as mgn  # noqa: E402  (seeded input helpers only; nothing reads /root/reference here)
from oracle import net as onet  # noqa: E402
from oracle import tiling  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_network.npz"))


@pytest.mark.parametrize("name", ["small", "vitl"])
def test_oracle_attention_equals_reference_flash_forward(gold, name):
    dim, heads, hw, rh, rw, B, seed, stride = (int(v) for v in gold[f"ff_{name}_cfg"])
    x, p = mgn.attention_case(dim, heads, hw, rh, rw, B, seed)
    assert mgn.checksum(x) == pytest.approx(float(gold[f"ff_{name}_xsum"]), rel=1e-12)     # RNG drift guard
    sd = {"a." + k: v for k, v in p.items()}
    with torch.no_grad():
        y = onet._attention(sd, "a.", x, heads).reshape(B, hw * hw, dim)[:, ::stride].numpy()
    ref = gold[f"ff_{name}_y"]
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_oracle_forward_equals_reference_class_transformer_forward(gold):
    embed, depth, ncls, tokens, seed = (int(v) for v in gold["ct_cfg"])
    sd = mgn.small_transformer_state(embed, depth, ncls, tokens, seed)
    xin = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(seed + 1))
    assert mgn.checksum(xin) == pytest.approx(float(gold["ct_xsum"]), rel=1e-12)
    y = onet.class_transformer_forward(sd, xin).numpy()
    ref = gold["ct_y"]
    assert y.shape == ref.shape == (2, ncls + 3, 64, 64)          # [class logits | dY dX cellprob], vit_sam.py:193
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()


def test_oracle_run_net_equals_reference_run_net(gold):
    for k in range(int(gold["rn_n"])):
        H, W, aug, ncls, bs, seed = (int(v) for v in gold[f"rn_{k}_cfg"])
        x = tiling.normalize_img(mgn.run_net_tile(H, W, seed)[None])
        assert mgn.checksum(x) == pytest.approx(float(gold[f"rn_{k}_xsum"]), rel=1e-9)

        def fw(img):
            o = mgn.fake_net_outputs(torch.from_numpy(np.ascontiguousarray(img)), ncls).numpy()
            return o[:, ncls:], o[:, :ncls]                       # the channel split of core._forward (core.py:69-71)
        dP, cp, yc = tiling.run_net(fw, x, batch_size=bs, augment=bool(aug), tile_overlap=0.1, bsize=256)
        yf = np.concatenate([dP, cp[None]], 0).transpose(1, 2, 0)     # (H, W, 3) like core.run_net returns
        assert np.array_equal(yf[::5, ::5], gold[f"rn_{k}_yf"]), k
        assert np.array_equal(yc.transpose(1, 2, 0)[::5, ::5], gold[f"rn_{k}_ycf"]), k
        assert np.array_equal(yf[H // 2], gold[f"rn_{k}_yf_row"]), k


def test_oracle_tile_pipeline_equals_reference_eval():
    """The per-tile orchestration: ``ClassposeModel.eval([tile], batch_size, augment, bsize=256, compute_masks=True)`` as
    the WSI worker calls it (predict_wsi.py:751-757), run by the reference itself on an elementwise stand-in network
    (tests/golden/make_golden_eval.py), against the composition the GPU tests use as their oracle: normalize_img ->
    run_net -> compute_masks -> compute_class_masks with the default arguments.  Bit for bit, and the arguments the
    reference hands across the cellpose boundary are the ones the oracle / the engine hard-code."""
    import json
    import make_golden_eval as mge
    from oracle import classmask, dynamics
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_eval.npz"))
    for k in range(int(g["ev_n"])):
        seed, tta, bs = (int(v) for v in g[f"ev_{k}_cfg"])
        tile = mge.eval_tile(seed)
        assert int(tile.astype(np.int64).sum()) == int(g[f"ev_{k}_tilesum"])        # RNG drift guard
        x = tiling.normalize_img(tile[None])

        def fw(img):
            o = mge.decode_net(torch.from_numpy(np.ascontiguousarray(img)), mge.NCLS).numpy()
            return o[:, mge.NCLS:], o[:, :mge.NCLS]
        dP, cp, yc = tiling.run_net(fw, x, batch_size=bs, augment=bool(tta), tile_overlap=0.1, bsize=256)
        assert np.array_equal(dP[:, ::4, ::4], g[f"ev_{k}_dP"]) and np.array_equal(cp[::4, ::4], g[f"ev_{k}_cellprob"])
        assert np.array_equal(yc[:, ::4, ::4], g[f"ev_{k}_yclass"])
        masks = dynamics.compute_masks(dP, cp)                                        # niter 200, 0.0 / 0.4, min_size 15, 0.4
        assert masks.max() > 50
        assert np.array_equal(masks.astype(np.uint16), g[f"ev_{k}_masks"])
        cm, _ = classmask.compute_class_masks(masks, yc)
        assert np.array_equal(cm.astype(np.uint8), g[f"ev_{k}_class_masks"])
        log = json.loads(str(g[f"ev_{k}_log"]))
        assert log["resize_and_compute_masks"] == {"cellprob_threshold": 0.0, "flow_threshold": 0.4, "max_size_fraction": 0.4,
                                                   "min_size": 15, "niter": 200, "resize": None, "dP_shape": [2, 256, 256],
                                                   "device": "device(type='cpu')"}
        n = log["normalize_img"]
        assert n["normalize"] is True and n["invert"] is False and n["percentile"] is None and n["lowhigh"] is None
        assert n["sharpen_radius"] == 0 and n["smooth_radius"] == 0 and n["tile_norm_blocksize"] == 0
        assert log["convert_image"]["shape"] == [256, 256, 3] and log["convert_image"]["do_3D"] is False
        assert list(g[f"ev_{k}_shape_x"]) == [1, 256, 256, 3] and int(g[f"ev_{k}_n_flows"]) == 5
