"""Mint golden vectors from the reference's OWN network glue (build container only):

* ``classpose.vit_sam.flash_forward``          /root/reference/src/classpose/vit_sam.py:15-65
* ``classpose.vit_sam.ClassTransformer.forward``   vit_sam.py:148-197
* ``classpose.core.run_net`` / ``_forward``    /root/reference/src/classpose/core.py:51-231

These are Classpose-owned functions; they import under the stub finder of make_golden.py.  What they
call in the ABSENT wheels is supplied as data-free stand-ins at the call boundary only:
``segment_anything...get_rel_pos`` and ``cellpose.transforms.{get_pad_yx, make_tiles, average_tiles,
unaugment_tiles}`` are bound to the oracle's restatements (so those stay "unpinned"), and the SAM
``Block`` / ``Attention`` / ``MLPBlock`` / ``LayerNorm2d`` containers that ``ClassTransformer.forward``
iterates over are plain ``torch.nn`` modules built here with the published layer structure.  What the
vectors PIN is everything the reference itself owns: the qkv reshape / head split, SDPA-with-bias
semantics and scale, the einsum of the decomposed rel-pos bias, the residual / block order, W2 / W3
pixel shuffles and the ``cat((x2, x1))`` channel order, the channel split of ``_forward``, and the
batching / un-augment / average / crop control flow of ``run_net``.

Inputs are regenerated from seeds by the tests (a checksum of each regenerated input is stored so RNG
drift is detected instead of mis-reported as a parity failure); outputs are stored, sub-sampled where
large.  Fixtures hold data only.        python tests/golden/make_golden_network.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402


# ---- seeded inputs shared with the tests (tests import these helpers) -------------------------------------
def attention_case(dim: int, heads: int, hw: int, rows_h: int, rows_w: int, batch: int, seed: int):
    """seeded (x, parameters) of one SAM attention module; float32"""
    g = torch.Generator().manual_seed(seed)
    p = {
        "qkv.weight": torch.randn(3 * dim, dim, generator=g) / dim ** 0.5,
        "qkv.bias": torch.randn(3 * dim, generator=g) * 0.1,
        "proj.weight": torch.randn(dim, dim, generator=g) / dim ** 0.5,
        "proj.bias": torch.randn(dim, generator=g) * 0.1,
        "rel_pos_h": torch.randn(rows_h, dim // heads, generator=g) * 0.2,
        "rel_pos_w": torch.randn(rows_w, dim // heads, generator=g) * 0.2,
    }
    x = torch.randn(batch, hw, hw, dim, generator=g)
    return x, p


def fake_net_outputs(X: torch.Tensor, ncls: int) -> torch.Tensor:
    """Deterministic elementwise stand-in for a network: (B, 3, b, b) -> (B, ncls + 3, b, b), class channels
    first like ClassTransformer.forward.  Flip-sensitive (row / column ramps) so that un-augmenting matters;
    only +, -, * on float32 so every CPU gives the same bits."""
    B, _, h, w = X.shape
    ry = torch.arange(h, dtype=torch.float32).reshape(1, h, 1) * 0.001
    rx = torch.arange(w, dtype=torch.float32).reshape(1, 1, w) * 0.002
    chans = []
    for c in range(ncls):
        chans.append(X[:, c % 3] * (0.5 + 0.25 * c) - X[:, (c + 1) % 3] * 0.125 + ry * (c + 1) - rx)
    chans.append(X[:, 0] - X[:, 1] * 0.5 + ry)              # dY
    chans.append(X[:, 2] * 2.0 + X[:, 0] - rx)              # dX
    chans.append(X[:, 1] * X[:, 2] + ry * rx * 10.0)        # cellprob
    return torch.stack(chans, 1)


def run_net_tile(H: int, W: int, seed: int) -> np.ndarray:
    """seeded uint8 RGB tile; run_net's input is its normalised image (models.py:642-666 runs normalize_img first)"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H // 8 + 1, W // 8 + 1, 3))
    tile = np.kron(base, np.ones((8, 8, 1), np.int64))[:H, :W] + rng.integers(-20, 21, (H, W, 3))
    return np.clip(tile, 0, 255).astype(np.uint8)


def small_transformer_state(embed: int, depth: int, ncls: int, tokens: int, seed: int) -> dict:
    """seeded state dict with the reference's key layout for a reduced ClassTransformer (head_dim 64)"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: torch.randn(*s, generator=g) * k
    sd = {"encoder.patch_embed.proj.weight": r(embed, 3, 8, 8, k=0.1), "encoder.patch_embed.proj.bias": r(embed, k=0.1),
          "encoder.pos_embed": r(1, tokens, tokens, embed, k=0.1)}
    for i in range(depth):
        p = f"encoder.blocks.{i}."
        sd[p + "norm1.weight"] = 1 + r(embed, k=0.1); sd[p + "norm1.bias"] = r(embed, k=0.1)
        sd[p + "attn.qkv.weight"] = r(3 * embed, embed, k=embed ** -0.5); sd[p + "attn.qkv.bias"] = r(3 * embed, k=0.1)
        sd[p + "attn.proj.weight"] = r(embed, embed, k=embed ** -0.5); sd[p + "attn.proj.bias"] = r(embed, k=0.1)
        sd[p + "attn.rel_pos_h"] = r(2 * tokens - 1 if i % 2 == 0 else 27, 64, k=0.2)    # odd layers: interpolated table
        sd[p + "attn.rel_pos_w"] = r(2 * tokens - 1 if i % 2 == 0 else 27, 64, k=0.2)
        sd[p + "norm2.weight"] = 1 + r(embed, k=0.1); sd[p + "norm2.bias"] = r(embed, k=0.1)
        sd[p + "mlp.lin1.weight"] = r(4 * embed, embed, k=embed ** -0.5); sd[p + "mlp.lin1.bias"] = r(4 * embed, k=0.1)
        sd[p + "mlp.lin2.weight"] = r(embed, 4 * embed, k=(4 * embed) ** -0.5); sd[p + "mlp.lin2.bias"] = r(embed, k=0.1)
    sd["encoder.neck.0.weight"] = r(256, embed, 1, 1, k=embed ** -0.5)
    sd["encoder.neck.1.weight"] = 1 + r(256, k=0.1); sd["encoder.neck.1.bias"] = r(256, k=0.1)
    sd["encoder.neck.2.weight"] = r(256, 256, 3, 3, k=(9 * 256) ** -0.5)
    sd["encoder.neck.3.weight"] = 1 + r(256, k=0.1); sd["encoder.neck.3.bias"] = r(256, k=0.1)
    sd["out.weight"] = r(192, 256, 1, 1, k=1 / 16); sd["out.bias"] = r(192, k=0.1)
    sd["W2"] = torch.eye(192).reshape(192, 3, 8, 8)
    sd["out_class.weight"] = r(ncls * 64, 256, 1, 1, k=1 / 16); sd["out_class.bias"] = r(ncls * 64, k=0.1)
    sd["W3"] = torch.eye(ncls * 64).reshape(ncls * 64, ncls, 8, 8)
    return sd


def checksum(t) -> float:
    a = t.detach().double().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, np.float64)
    return float((a * np.cos(np.arange(a.size, dtype=np.float64).reshape(a.shape) * 0.37)).sum())


# ---- published container structure of segment_anything 1.0 (layer wiring only) ----------------------------
class Attention(nn.Module):          # the class NAME is what patch_attention_forwards matches (vit_sam.py:71)
    def __init__(self, dim, heads, rows_h, rows_w):
        super().__init__()
        self.num_heads, self.scale, self.use_rel_pos = heads, (dim // heads) ** -0.5, True
        self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(rows_h, dim // heads))
        self.rel_pos_w = nn.Parameter(torch.zeros(rows_w, dim // heads))


class MLPBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lin1, self.lin2, self.act = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim), nn.GELU()

    def forward(self, x):
        return self.lin2(self.act(self.lin1(x)))


class Block(nn.Module):              # window_size == 0 (cellpose sets global attention in every block)
    def __init__(self, dim, heads, rows_h, rows_w):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn, self.mlp = Attention(dim, heads, rows_h, rows_w), MLPBlock(dim)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class LayerNorm2d(nn.Module):
    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c)), eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class PatchEmbed(nn.Module):
    def __init__(self, embed):
        super().__init__()
        self.proj = nn.Conv2d(3, embed, 8, 8)

    def forward(self, x):
        return self.proj(x).permute(0, 2, 3, 1)


def main():
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, mg.REF)
    from oracle import net as onet
    from oracle import tiling
    # stand-ins at the wheel boundary (see module docstring)
    import segment_anything.modeling.image_encoder as ie
    ie.get_rel_pos = onet.get_rel_pos
    import cellpose.transforms as ctf
    ctf.get_pad_yx, ctf.make_tiles = tiling.get_pad_yx, tiling.make_tiles
    ctf.average_tiles, ctf.unaugment_tiles = tiling.average_tiles, tiling.unaugment_tiles
    import cellpose.core as ccore
    ccore.tqdm_out = None
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm")).trange = range
    from classpose import core as rcore
    from classpose import vit_sam as rvit

    out = {}
    torch.manual_seed(0)

    # ---- flash_forward: small (oracle check, rel-pos table interpolated on w) and ViT-L sized (HIP check)
    for name, (dim, heads, hw, rh, rw, B, seed, stride) in {
            "small": (128, 2, 8, 15, 27, 2, 101, 1), "vitl": (1024, 16, 32, 63, 127, 2, 102, 32)}.items():
        x, p = attention_case(dim, heads, hw, rh, rw, B, seed)
        m = Attention(dim, heads, rh, rw)
        m.load_state_dict(p)
        with torch.no_grad():
            y = rvit.flash_forward(m, x)
        out[f"ff_{name}_cfg"] = np.array([dim, heads, hw, rh, rw, B, seed, stride])
        out[f"ff_{name}_xsum"] = np.array(checksum(x))
        out[f"ff_{name}_y"] = y.reshape(B, hw * hw, dim)[:, ::stride].numpy()

    # ---- ClassTransformer.forward on a reduced model (embed 128 = 2 heads of 64, depth 2, 64 px -> 8 x 8 tokens)
    embed, depth, ncls, tokens, seed = 128, 2, 3, 8, 103
    sd = small_transformer_state(embed, depth, ncls, tokens, seed)
    net = rvit.ClassTransformer.__new__(rvit.ClassTransformer)
    nn.Module.__init__(net)
    enc = nn.Module()
    enc.patch_embed = PatchEmbed(embed)
    enc.pos_embed = nn.Parameter(torch.zeros(1, tokens, tokens, embed))
    enc.blocks = nn.ModuleList([Block(embed, embed // 64, sd[f"encoder.blocks.{i}.attn.rel_pos_h"].shape[0],
                                      sd[f"encoder.blocks.{i}.attn.rel_pos_w"].shape[0]) for i in range(depth)])
    enc.neck = nn.Sequential(nn.Conv2d(embed, 256, 1, bias=False), LayerNorm2d(256),
                             nn.Conv2d(256, 256, 3, padding=1, bias=False), LayerNorm2d(256))
    net.encoder = enc
    net.out = nn.Conv2d(256, 192, 1)
    net.W2 = nn.Parameter(torch.zeros(192, 3, 8, 8), requires_grad=False)
    net.out_class = nn.Conv2d(256, ncls * 64, 1)
    net.W3 = nn.Parameter(torch.zeros(ncls * 64, ncls, 8, 8), requires_grad=False)
    net.ps, net.n_cell_classes, net.rdrop = 8, ncls, 0.0
    net.load_state_dict(sd)
    rvit.patch_attention_forwards(net)                      # the reference's own patching of Attention.forward
    net.eval()
    g = torch.Generator().manual_seed(seed + 1)
    xin = torch.rand(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        y, _style = net(xin)
    out["ct_cfg"] = np.array([embed, depth, ncls, tokens, seed])
    out["ct_xsum"] = np.array(checksum(xin))
    out["ct_y"] = y.numpy()

    # ---- core.run_net + _forward with the elementwise fake network
    class FakeNet(nn.Module):
        def __init__(self, ncls):
            super().__init__()
            self.n_cell_classes, self.device = ncls, torch.device("cpu")
            self.dummy = nn.Parameter(torch.zeros(1))

        def forward(self, X):
            return fake_net_outputs(X, self.n_cell_classes), torch.zeros(X.shape[0], 256)

    cases = [(256, 256, False, 3, 8, 201), (256, 256, True, 3, 4, 202), (320, 288, True, 2, 8, 203), (512, 512, False, 2, 8, 204)]
    for k, (H, W, aug, ncls_, bs, seed_) in enumerate(cases):
        x = tiling.normalize_img(run_net_tile(H, W, seed_)[None])
        yf, ycf, _ = rcore.run_net(FakeNet(ncls_), x, batch_size=bs, augment=aug, tile_overlap=0.1, bsize=256)
        out[f"rn_{k}_cfg"] = np.array([H, W, int(aug), ncls_, bs, seed_])
        out[f"rn_{k}_xsum"] = np.array(checksum(x))
        out[f"rn_{k}_yf"] = yf[0, ::5, ::5].copy()          # (H/5, W/5, 3) sub-sample; edges + interior covered
        out[f"rn_{k}_ycf"] = ycf[0, ::5, ::5].copy()
        out[f"rn_{k}_yf_row"] = yf[0, H // 2].copy()        # one full row, every column (taper seams)
    out["rn_n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "reference_network.npz"), **out)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "reference_network.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
