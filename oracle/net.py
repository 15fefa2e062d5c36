"""Oracle (test infrastructure): ClassTransformer forward on torch-CPU.

Restates, as a functional forward over a plain state dict,

* ``ClassTransformer.forward``  /root/reference/src/classpose/vit_sam.py:148-197
* ``flash_forward``             /root/reference/src/classpose/vit_sam.py:15-65
* ``UNet.forward``              /root/reference/src/classpose/unet.py:173-196
* cellpose==4.0.8 ``vit_sam.Transformer.__init__`` and segment-anything==1.0
  ``ImageEncoderViT / Block / Attention / MLPBlock / LayerNorm2d / get_rel_pos``
  (third-party, absent from /root/reference).  PARITY
  UNPINNED for those (see oracle/__init__.py); the Classpose-owned parts (head,
  W3 pixel shuffle, channel order, SDPA-with-bias semantics) follow the in-tree
  file line by line and the UNet head is pinned by a golden vector generated
  from the reference's own ``classpose.unet.UNet``.

State-dict key layout = what ``net.load_model`` / ``infer_structure`` expect
(predict_wsi.py:1393-1405).  Not imported by anything under ``classpose_amd/``.
"""
from __future__ import annotations

import re

import torch
import torch.nn.functional as F

PS = 8          # patch size (cellpose Transformer ps)
NOUT = 3        # dY, dX, cellprob
EMBED = 1024    # vit_l
HEADS = 16
DEPTH = 24
MLP = 4096
NECK = 256
GLOBAL_ATTN = (5, 11, 17, 23)   # SAM vit_l: rel_pos tables 127x64 there, 27x64 elsewhere


def get_rel_pos(q_size: int, k_size: int, rel_pos: torch.Tensor) -> torch.Tensor:
    """segment_anything.modeling.image_encoder.get_rel_pos (call site vit_sam.py:40-41)."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        rel_pos_resized = F.interpolate(
            rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1),
            size=max_rel_dist, mode="linear")
        rel_pos_resized = rel_pos_resized.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        rel_pos_resized = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    relative_coords = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos_resized[relative_coords.long()]


def _attention(sd, pfx: str, x: torch.Tensor, num_heads: int) -> torch.Tensor:
    """flash_forward, vit_sam.py:26-65."""
    B, H, W, C = x.shape
    L = H * W
    qkv = F.linear(x, sd[pfx + "qkv.weight"], sd[pfx + "qkv.bias"])
    qkv = qkv.reshape(B, L, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    head_dim = q.shape[-1]
    scale = head_dim ** -0.5
    q_hw = q.reshape(B, num_heads, H, W, head_dim)
    Rh = get_rel_pos(H, H, sd[pfx + "rel_pos_h"])
    Rw = get_rel_pos(W, W, sd[pfx + "rel_pos_w"])
    rel_h = torch.einsum("b n h w c, h k c -> b n h w k", q_hw, Rh)
    rel_w = torch.einsum("b n h w c, w k c -> b n h w k", q_hw, Rw)
    bias = (rel_h[..., :, None] + rel_w[..., None, :]).reshape(B, num_heads, L, L)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, dropout_p=0.0,
                                       is_causal=False, scale=scale)
    o = o.transpose(1, 2).reshape(B, H, W, -1)
    return F.linear(o, sd[pfx + "proj.weight"], sd[pfx + "proj.bias"])


def _layernorm2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def _unet_block(sd, pfx, x, skip_last_activation=False):
    x = F.conv2d(x, sd[pfx + "conv1.weight"], sd[pfx + "conv1.bias"], padding=1)
    x = F.relu(x)
    x = F.conv2d(x, sd[pfx + "conv2.weight"], sd[pfx + "conv2.bias"], padding=1)
    if not skip_last_activation:
        x = F.relu(x)
    return x


def unet_forward(sd, pfx: str, x: torch.Tensor, n_enc: int) -> torch.Tensor:
    """unet.py:173-196 over state-dict keys ``<pfx>encoder_blocks.N...``."""
    feats = []
    for i in range(n_enc):
        p = f"{pfx}encoder_blocks.{i}."
        x = _unet_block(sd, p + "block.", x)
        x = F.conv2d(x, sd[p + "downconv.weight"], sd[p + "downconv.bias"], stride=2)
        feats.append(x)
    feats = feats[::-1]
    p = pfx + "bottleneck_down."
    x = _unet_block(sd, p + "block.", x)
    x = F.conv2d(x, sd[p + "downconv.weight"], sd[p + "downconv.bias"], stride=2)
    p = pfx + "bottleneck_up."
    x = _unet_block(sd, p + "block.", x)
    x = F.conv_transpose2d(x, sd[p + "upconv.weight"], sd[p + "upconv.bias"], stride=2)
    for i in range(n_enc):
        p = f"{pfx}decoder_blocks.{i}."
        x = _unet_block(sd, p + "block.", torch.cat((x, feats[i]), dim=1),
                        skip_last_activation=(i == n_enc - 1))
        x = F.conv_transpose2d(x, sd[p + "upconv.weight"], sd[p + "upconv.bias"], stride=2)
    return x


def infer_structure(sd) -> tuple[list[int] | None, int, int]:
    """predict_wsi.py:1377-1419 (+ depth, for reduced test models)."""
    fts = [sd[k].shape[0] for k in sd
           if re.search(r"out_class\.encoder_blocks\.[0-9]+\.block.conv1.weight", k)]
    n_classes = sd["W3"].shape[1]
    depth = 1 + max(int(m.group(1)) for k in sd
                    if (m := re.match(r"encoder\.blocks\.(\d+)\.norm1\.weight", k)))
    return (fts or None), n_classes, depth


@torch.no_grad()
def class_transformer_forward(sd: dict, x: torch.Tensor, dtype=torch.float32,
                              return_tokens: bool = False):
    """x (B,3,256,256) float32 -> out (B, ncls+3, 256, 256) float32.

    Mirrors core._forward (core.py:61-68: cast to net dtype, forward, cast back
    to float32) + ClassTransformer.forward.  ``sd`` must already be in ``dtype``.
    """
    fts, ncls, depth = infer_structure(sd)
    heads = sd["encoder.blocks.0.attn.qkv.weight"].shape[1] // 64   # head_dim 64 (vit_l: 16)
    x = x.to(dtype)
    x = F.conv2d(x, sd["encoder.patch_embed.proj.weight"],
                 sd["encoder.patch_embed.proj.bias"], stride=PS)
    x = x.permute(0, 2, 3, 1)
    x = x + sd["encoder.pos_embed"]
    for i in range(depth):
        p = f"encoder.blocks.{i}."
        h = F.layer_norm(x, x.shape[-1:], sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        x = x + _attention(sd, p + "attn.", h, heads)
        h = F.layer_norm(x, x.shape[-1:], sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = F.linear(h, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])
        h = F.gelu(h)
        x = x + F.linear(h, sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
    tokens = x
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd["encoder.neck.0.weight"])
    x = _layernorm2d(x, sd["encoder.neck.1.weight"], sd["encoder.neck.1.bias"])
    x = F.conv2d(x, sd["encoder.neck.2.weight"], padding=1)
    x = _layernorm2d(x, sd["encoder.neck.3.weight"], sd["encoder.neck.3.bias"])
    x1 = F.conv2d(x, sd["out.weight"], sd["out.bias"])
    x1 = F.conv_transpose2d(x1, sd["W2"], stride=PS, padding=0)
    if fts is not None:
        x2 = unet_forward(sd, "out_class.", x, len(fts))
    else:
        x2 = F.conv2d(x, sd["out_class.weight"], sd["out_class.bias"])
    x2 = F.conv_transpose2d(x2, sd["W3"], stride=PS, padding=0)
    out = torch.cat((x2, x1), 1).float()
    if return_tokens:
        return out, tokens.float(), x.float()
    return out


def make_forward(sd: dict, dtype=torch.float32):
    """Adapter for oracle.tiling.run_net: numpy in, (y[B,3], y_class[B,ncls]) out
    (channel split of core.py:69-71)."""
    import numpy as np
    sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    ncls = sd["W3"].shape[1]

    def forward(img: "np.ndarray"):
        out = class_transformer_forward(sdd, torch.from_numpy(np.ascontiguousarray(img)), dtype)
        out = out.numpy()
        return out[:, ncls:], out[:, :ncls]
    return forward
