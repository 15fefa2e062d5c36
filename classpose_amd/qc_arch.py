"""Architecture tables of the GrandQC networks (tissue: 2 classes, artefacts: 8 classes).

The reference builds them with ``smp.UnetPlusPlus(encoder_name="timm-efficientnet-b0",
classes=...)`` (/root/reference/src/classpose/grandqc/wsi_tissue_detection.py:86-91; the artefact
model is a pickled module of the same family, wsi_artefact_detection.py:124).  Both libraries are
third-party dependencies that are absent from this image (segmentation-models-pytorch 0.3.1,
timm 0.4.12 in the reference's lock file): this module restates their published layer tables --
EfficientNet-B0 ('ds_r1_k3_s1_e1_c16_se0.25' ... 'ir_r1_k3_s1_e6_c320_se0.25', symmetric padding,
BatchNorm eps 1e-5, Swish, SE reduction = 0.25 x block input channels) and the UNet++ decoder
(decoder_channels 256,128,64,32,16; nearest x2 upsampling; Conv3x3-BN-ReLU pairs; dense skips)
-- together with the state-dict key names those libraries produce, so that a real checkpoint
(``torch.save(model.state_dict())``) loads by name and every shape is checked.
PARITY UNPINNED against smp/timm themselves (not importable here).
"""
from __future__ import annotations

BN_EPS = 1e-5
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # smp get_preprocessing_fn("timm-efficientnet-b0", "imagenet")
IMAGENET_STD = (0.229, 0.224, 0.225)
STEM = 32
# (block type, repeats, kernel, stride, expansion, out channels) per stage
STAGES = [("ds", 1, 3, 1, 1, 16), ("ir", 2, 3, 2, 6, 24), ("ir", 2, 5, 2, 6, 40), ("ir", 3, 3, 2, 6, 80),
          ("ir", 3, 5, 1, 6, 112), ("ir", 4, 5, 2, 6, 192), ("ir", 1, 3, 1, 6, 320)]
# encoder features handed to the decoder: after stem, after stages 1, 2, 4, 6 (smp stage_idxs (2, 3, 5))
FEATURE_AFTER_STAGE = {1: 1, 2: 2, 4: 3, 6: 4}       # stage index -> feature slot (0 = stem)
ENCODER_CHANNELS = (3, 32, 24, 40, 112, 320)
DECODER_CHANNELS = (256, 128, 64, 32, 16)


def encoder_blocks():
    """Flat list of dicts: one per MBConv block in execution order."""
    out = []
    cin = STEM
    for s, (kind, rep, k, stride, e, cout) in enumerate(STAGES):
        for b in range(rep):
            st = stride if b == 0 else 1
            out.append(dict(stage=s, block=b, kind=kind, k=k, stride=st, cin=cin, mid=cin * e, cout=cout,
                            se=max(1, int(cin * 0.25)), residual=(st == 1 and cin == cout),
                            prefix=f"encoder.blocks.{s}.{b}."))
            cin = cout
    return out


def decoder_blocks():
    """UNet++ decoder blocks: name -> (in_channels, skip_channels, out_channels)."""
    enc = list(ENCODER_CHANNELS[1:])[::-1]                 # 320, 112, 40, 24, 32
    in_ch = [enc[0]] + list(DECODER_CHANNELS[:-1])         # 320, 256, 128, 64, 32
    skip_ch = enc[1:] + [0]                                # 112, 40, 24, 32, 0
    out_ch = list(DECODER_CHANNELS)
    blocks = {}
    for layer in range(len(in_ch) - 1):
        for depth in range(layer + 1):
            if depth == 0:
                blocks[f"x_{depth}_{layer}"] = (in_ch[layer], skip_ch[layer] * (layer + 1), out_ch[layer])
            else:
                blocks[f"x_{depth}_{layer}"] = (skip_ch[layer - 1], skip_ch[layer] * (layer + 1 - depth),
                                                skip_ch[layer])
    blocks[f"x_0_{len(in_ch) - 1}"] = (in_ch[-1], 0, out_ch[-1])
    return blocks


def decoder_schedule():
    """Execution order of the dense decoder: (block name, x source, [skip sources]).  Sources are
    'f<k>' (reversed encoder features: f0 = 320 ch at stride 32 ... f4 = 32 ch at stride 2) or
    block names."""
    depth = 4
    sched = []
    for layer in range(depth):
        for d in range(depth - layer):
            if layer == 0:
                sched.append((f"x_{d}_{d}", f"f{d}", [f"f{d + 1}"]))
            else:
                li = d + layer
                cat = [f"x_{i}_{li}" for i in range(d + 1, li + 1)] + [f"f{li + 1}"]
                sched.append((f"x_{d}_{li}", f"x_{d}_{li - 1}", cat))
    sched.append((f"x_0_{depth}", f"x_0_{depth - 1}", []))
    return sched


def expected_shapes(n_classes: int) -> dict[str, tuple]:
    """state-dict key -> shape for every tensor the forward pass reads."""
    sh = {"encoder.conv_stem.weight": (STEM, 3, 3, 3)}

    def bn(p, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            sh[p + "." + n] = (c,)

    bn("encoder.bn1", STEM)
    for b in encoder_blocks():
        p = b["prefix"]
        if b["kind"] == "ds":
            sh[p + "conv_dw.weight"] = (b["cin"], 1, b["k"], b["k"]); bn(p + "bn1", b["cin"])
            sh[p + "se.conv_reduce.weight"] = (b["se"], b["cin"], 1, 1); sh[p + "se.conv_reduce.bias"] = (b["se"],)
            sh[p + "se.conv_expand.weight"] = (b["cin"], b["se"], 1, 1); sh[p + "se.conv_expand.bias"] = (b["cin"],)
            sh[p + "conv_pw.weight"] = (b["cout"], b["cin"], 1, 1); bn(p + "bn2", b["cout"])
        else:
            m = b["mid"]
            sh[p + "conv_pw.weight"] = (m, b["cin"], 1, 1); bn(p + "bn1", m)
            sh[p + "conv_dw.weight"] = (m, 1, b["k"], b["k"]); bn(p + "bn2", m)
            sh[p + "se.conv_reduce.weight"] = (b["se"], m, 1, 1); sh[p + "se.conv_reduce.bias"] = (b["se"],)
            sh[p + "se.conv_expand.weight"] = (m, b["se"], 1, 1); sh[p + "se.conv_expand.bias"] = (m,)
            sh[p + "conv_pwl.weight"] = (b["cout"], m, 1, 1); bn(p + "bn3", b["cout"])
    for name, (ci, cs, co) in decoder_blocks().items():
        p = f"decoder.blocks.{name}."
        sh[p + "conv1.0.weight"] = (co, ci + cs, 3, 3); bn(p + "conv1.1", co)
        sh[p + "conv2.0.weight"] = (co, co, 3, 3); bn(p + "conv2.1", co)
    sh["segmentation_head.0.weight"] = (n_classes, DECODER_CHANNELS[-1], 3, 3)
    sh["segmentation_head.0.bias"] = (n_classes,)
    return sh


def check_state_dict(sd: dict) -> int:
    """Validates names/shapes against the tables; returns the number of classes."""
    if "segmentation_head.0.weight" not in sd:
        raise ValueError("not a GrandQC UNet++ state dict: segmentation_head.0.weight missing")
    n_classes = int(sd["segmentation_head.0.weight"].shape[0])
    for k, shape in expected_shapes(n_classes).items():
        if k not in sd:
            raise ValueError(f"GrandQC state dict misses {k}")
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"GrandQC state dict: {k} has shape {tuple(sd[k].shape)}, expected {shape}")
    return n_classes
