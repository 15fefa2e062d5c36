"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32) restatement of the GrandQC networks.

``smp.UnetPlusPlus("timm-efficientnet-b0")`` forward as the reference runs it through
``model.predict`` (/root/reference/src/classpose/grandqc/wsi_tissue_detection.py:86-160,
wsi_artefact_detection.py:124-195): ImageNet preprocessing, EfficientNet-B0 encoder, UNet++
decoder, 3x3 segmentation head, argmax.  smp 0.3.1 / timm 0.4.12 are absent from the image:
the layer tables live in classpose_amd/qc_arch.py (a pure table module, no compute) and this
file applies them with plain torch.nn.functional calls.  PARITY UNPINNED (see qc_arch.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F



class A:
    """The oracle's own copy of the layer tables, written out literally (hand-derived from the
    smp 0.3.1 / timm 0.4.12 sources' published structure) so that the product's generated tables
    (classpose_amd/qc_arch.py) are cross-checked against an independent statement
    (tests/test_qc_host.py::test_arch_tables_agree)."""
    BN_EPS = 1e-5
    IMAGENET_MEAN = (0.485, 0.456, 0.406)
    IMAGENET_STD = (0.229, 0.224, 0.225)
    # (stage, block, kind, k, stride, cin, mid, cout, se, residual)
    ENC = [(0, 0, "ds", 3, 1, 32, 32, 16, 8, False),
           (1, 0, "ir", 3, 2, 16, 96, 24, 4, False), (1, 1, "ir", 3, 1, 24, 144, 24, 6, True),
           (2, 0, "ir", 5, 2, 24, 144, 40, 6, False), (2, 1, "ir", 5, 1, 40, 240, 40, 10, True),
           (3, 0, "ir", 3, 2, 40, 240, 80, 10, False), (3, 1, "ir", 3, 1, 80, 480, 80, 20, True),
           (3, 2, "ir", 3, 1, 80, 480, 80, 20, True),
           (4, 0, "ir", 5, 1, 80, 480, 112, 20, False), (4, 1, "ir", 5, 1, 112, 672, 112, 28, True),
           (4, 2, "ir", 5, 1, 112, 672, 112, 28, True),
           (5, 0, "ir", 5, 2, 112, 672, 192, 28, False), (5, 1, "ir", 5, 1, 192, 1152, 192, 48, True),
           (5, 2, "ir", 5, 1, 192, 1152, 192, 48, True), (5, 3, "ir", 5, 1, 192, 1152, 192, 48, True),
           (6, 0, "ir", 3, 1, 192, 1152, 320, 48, False)]
    FEATURE_AFTER_STAGE = {1: 1, 2: 2, 4: 3, 6: 4}
    SCHED = [("x_0_0", "f0", ["f1"]), ("x_1_1", "f1", ["f2"]), ("x_2_2", "f2", ["f3"]), ("x_3_3", "f3", ["f4"]),
             ("x_0_1", "x_0_0", ["x_1_1", "f2"]), ("x_1_2", "x_1_1", ["x_2_2", "f3"]),
             ("x_2_3", "x_2_2", ["x_3_3", "f4"]),
             ("x_0_2", "x_0_1", ["x_1_2", "x_2_2", "f3"]), ("x_1_3", "x_1_2", ["x_2_3", "x_3_3", "f4"]),
             ("x_0_3", "x_0_2", ["x_1_3", "x_2_3", "x_3_3", "f4"]),
             ("x_0_4", "x_0_3", [])]

    @staticmethod
    def encoder_blocks():
        keys = ("stage", "block", "kind", "k", "stride", "cin", "mid", "cout", "se", "residual")
        return [dict(zip(keys, e), prefix=f"encoder.blocks.{e[0]}.{e[1]}.") for e in A.ENC]

    @staticmethod
    def decoder_schedule():
        return A.SCHED


def preprocess(patch_u8: np.ndarray) -> torch.Tensor:
    """get_preprocessing (wsi_qc_helpers.py:104-120): (x/255 - mean)/std, HWC -> 1xCxHxW float32."""
    x = patch_u8.astype(np.float64) / 255.0          # smp preprocess_input works in float64 numpy
    x = (x - np.array(A.IMAGENET_MEAN)) / np.array(A.IMAGENET_STD)
    return torch.from_numpy(x.transpose(2, 0, 1).astype("float32"))[None]


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=A.BN_EPS)


def _swish(x):
    return x * torch.sigmoid(x)


def _se(sd, p, x):
    s = x.mean((2, 3), keepdim=True)
    s = _swish(F.conv2d(s, sd[p + "conv_reduce.weight"], sd[p + "conv_reduce.bias"]))
    s = F.conv2d(s, sd[p + "conv_expand.weight"], sd[p + "conv_expand.bias"])
    return x * torch.sigmoid(s)


def encoder_forward(sd, x):
    feats = [None] * 5
    x = _swish(_bn(sd, "encoder.bn1", F.conv2d(x, sd["encoder.conv_stem.weight"], stride=2, padding=1)))
    feats[0] = x
    blocks = A.encoder_blocks()
    for i, b in enumerate(blocks):
        p, k = b["prefix"], b["k"]
        inp = x
        if b["kind"] == "ds":
            x = _swish(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv_dw.weight"], stride=b["stride"], padding=k // 2,
                                                   groups=b["cin"])))
            x = _se(sd, p + "se.", x)
            x = _bn(sd, p + "bn2", F.conv2d(x, sd[p + "conv_pw.weight"]))
        else:
            x = _swish(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv_pw.weight"])))
            x = _swish(_bn(sd, p + "bn2", F.conv2d(x, sd[p + "conv_dw.weight"], stride=b["stride"], padding=k // 2,
                                                   groups=b["mid"])))
            x = _se(sd, p + "se.", x)
            x = _bn(sd, p + "bn3", F.conv2d(x, sd[p + "conv_pwl.weight"]))
        if b["residual"]:
            x = x + inp
        last_of_stage = i + 1 == len(blocks) or blocks[i + 1]["stage"] != b["stage"]
        if last_of_stage and b["stage"] in A.FEATURE_AFTER_STAGE:
            feats[A.FEATURE_AFTER_STAGE[b["stage"]]] = x
    return feats                                     # strides 2, 4, 8, 16, 32


def _decoder_block(sd, name, x, skip):
    p = f"decoder.blocks.{name}."
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    if skip is not None:
        x = torch.cat([x, skip], dim=1)
    x = F.relu(_bn(sd, p + "conv1.1", F.conv2d(x, sd[p + "conv1.0.weight"], padding=1)))
    x = F.relu(_bn(sd, p + "conv2.1", F.conv2d(x, sd[p + "conv2.0.weight"], padding=1)))
    return x


def forward(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """x: (n, 3, H, W) preprocessed float32, H and W multiples of 32 -> logits (n, classes, H, W)."""
    feats = encoder_forward(sd, x)
    t = {f"f{k}": feats[4 - k] for k in range(5)}     # f0 = deepest
    for name, xsrc, skips in A.decoder_schedule():
        skip = torch.cat([t[s] for s in skips], dim=1) if skips else None
        t[name] = _decoder_block(sd, name, t[xsrc], skip)
    return F.conv2d(t["x_0_4"], sd["segmentation_head.0.weight"], sd["segmentation_head.0.bias"], padding=1)


def predict_mask(sd: dict, patch_u8: np.ndarray) -> np.ndarray:
    """one 512x512x3 uint8 patch -> int8 class map, like the reference's per-patch body"""
    with torch.no_grad():
        pred = forward(sd, preprocess(patch_u8))[0].numpy()
    return np.argmax(pred, axis=0).astype("int8")


# ---- patch loops of the reference, restated literally (np.concatenate assembly) -------------------
def tissue_class_map_ref(image: np.ndarray, predict, p_s: int = 512) -> np.ndarray:
    """wsi_tissue_detection.py:113-196 with ``predict(patch_u8) -> int8 mask`` in place of the model."""
    from PIL import Image
    img = Image.fromarray(image)
    width, height = img.size
    wi_n, he_n = width // p_s, height // p_s
    overhang_wi, overhang_he = width - wi_n * p_s, height - he_n * p_s
    for h in range(he_n + 1):
        for w in range(wi_n + 1):
            if w != wi_n and h != he_n:
                box = (w * p_s, h * p_s, (w + 1) * p_s, (h + 1) * p_s)
            elif w == wi_n and h != he_n:
                box = (width - p_s, h * p_s, width, (h + 1) * p_s)
            elif w != wi_n and h == he_n:
                box = (w * p_s, height - p_s, (w + 1) * p_s, height)
            else:
                box = (width - p_s, height - p_s, width, height)
            mask = predict(np.array(img.crop(box)))
            if w == 0:
                temp = mask
            elif w == wi_n:
                temp = np.concatenate((temp, mask[:, p_s - overhang_wi: p_s]), axis=1)
            else:
                temp = np.concatenate((temp, mask), axis=1)
        if h == 0:
            end = temp
        elif h == he_n:
            end = np.concatenate((end, temp[p_s - overhang_he: p_s]), axis=0)
        else:
            end = np.concatenate((end, temp), axis=0)
    ah, aw = end.shape
    if (ah, aw) != (height, width):
        end = end[ah - height: ah, aw - width: aw]
    return end


def artefact_mask_ref(image: np.ndarray, tissue_mask_art: np.ndarray, predict, p_s: int = 512) -> np.ndarray:
    """wsi_artefact_detection.py:175-229"""
    height, width = image.shape[:2]
    n_w, n_h = width // p_s, height // p_s
    rows = []
    for h in range(n_h):
        cols = []
        for w in range(n_w):
            td = tissue_mask_art[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s]
            if np.count_nonzero(td == 1) > 50:
                raw = predict(image[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s])
                cols.append(np.where(td == 1, raw, 7))
            else:
                cols.append(np.full(td.shape, 7))
        rows.append(np.concatenate(cols, axis=1))
    out = np.concatenate(rows, axis=0)
    if height - n_h * p_s > 0:
        out = np.concatenate((out, np.full((height - n_h * p_s, out.shape[1]), 7, dtype=out.dtype)), axis=0)
    if width - n_w * p_s > 0:
        out = np.concatenate((out, np.full((out.shape[0], width - n_w * p_s), 7, dtype=out.dtype)), axis=1)
    return out
