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
