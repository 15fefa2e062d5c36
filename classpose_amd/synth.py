"""Procedural synthetic slide, analytic network fields and seeded weights.

There is no network in the build/bench environment, so neither a real slide nor
the reference's checkpoints (HF ``classpose/classpose``,
/root/reference/src/classpose/model_configs.py:23-109) can be fetched.  This
module provides

* ``SyntheticSlide`` -- an OpenSlide-protocol reader (the protocol ``WSIReader``
  hands to ``SlideLoader``, /root/reference/src/classpose/__init__.py:39-41 and
  wsi_utils.py:10-143) whose pixels are a pure function of (x, y, seed), so any
  region is reproducible on any rank without storage;
* ``analytic_fields`` -- flow / cellprob / class-logit tensors rendered from
  the same nuclei ("flow-injection" mode: random weights give no meaningful
  cells, so cells/s is measured on these while the network still runs);
* ``make_state_dict`` -- seeded random weights in the exact state-dict key
  layout ``net.load_model`` / ``infer_structure`` expect
  (/root/reference/src/classpose/entrypoints/predict_wsi.py:1393-1405,
  vit_sam.py:127-144, unet.py:146-171; cellpose/SAM layout SURVEY A.1).
"""
from __future__ import annotations

import numpy as np

PITCH = 28            # jittered-grid pitch (px): ~1.28e-3 nuclei / px^2
R_MIN, R_MAX = 5.0, 9.0
JITTER = 4.5          # |offset| <= JITTER  => centre distance >= 19 > 2*R_MAX
BG = np.array([230.0, 200.0, 220.0])
FG = np.array([90.0, 60.0, 140.0])
NOISE_SIGMA = 8.0


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _hash(seed: int, *keys) -> np.ndarray:
    with np.errstate(over="ignore"):
        h = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        for k in keys:
            h = _mix(h ^ (np.asarray(k).astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)))
    return h


def _u01(h: np.ndarray) -> np.ndarray:
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def nuclei_in_region(seed: int, x0: int, y0: int, w: int, h: int, margin: float = R_MAX + 1):
    """Centres (cx, cy), radii and ids of all nuclei that can touch the region."""
    gx0 = int(np.floor((x0 - margin - JITTER) / PITCH)) - 1
    gx1 = int(np.floor((x0 + w + margin + JITTER) / PITCH)) + 1
    gy0 = int(np.floor((y0 - margin - JITTER) / PITCH)) - 1
    gy1 = int(np.floor((y0 + h + margin + JITTER) / PITCH)) + 1
    gx, gy = np.meshgrid(np.arange(gx0, gx1 + 1), np.arange(gy0, gy1 + 1), indexing="xy")
    gx = gx.ravel().astype(np.int64)
    gy = gy.ravel().astype(np.int64)
    kx = gx + (1 << 20)
    ky = gy + (1 << 20)
    cx = (gx + 0.5) * PITCH + (_u01(_hash(seed, kx, ky, 1)) * 2 - 1) * JITTER
    cy = (gy + 0.5) * PITCH + (_u01(_hash(seed, kx, ky, 2)) * 2 - 1) * JITTER
    r = R_MIN + _u01(_hash(seed, kx, ky, 3)) * (R_MAX - R_MIN)
    ident = _hash(seed, kx, ky, 4)
    keep = (cx + r >= x0 - margin) & (cx - r <= x0 + w + margin) & \
           (cy + r >= y0 - margin) & (cy - r <= y0 + h + margin)
    return cx[keep], cy[keep], r[keep], ident[keep]


def _owner_map(seed, x0, y0, w, h, out_w=None, out_h=None):
    """Per sample: index of the nucleus containing the sample point, else -1.  Samples are
    the pixel centres of the region, or (out_w x out_h given) the centres of the pixels of
    the region rescaled to that size, in slide coordinates."""
    cx, cy, r, ident = nuclei_in_region(seed, x0, y0, w, h)
    ow, oh = (w, h) if out_w is None else (out_w, out_h)
    ys = (y0 + (np.arange(oh) + 0.5) * (h / oh) - 0.5)[:, None]
    xs = (x0 + (np.arange(ow) + 0.5) * (w / ow) - 0.5)[None, :]
    owner = np.full((oh, ow), -1, np.int64)
    if len(cx):
        # all nuclei at once: every disc's bounding window in sample indices, padded to the widest one (discs do not
        # overlap, so the write order is irrelevant); same float64 expressions as a per-nucleus loop
        ya = np.searchsorted(ys[:, 0], cy - r, side="left")
        yb = np.minimum(np.searchsorted(ys[:, 0], cy + r, side="left") + 1, oh)
        xa = np.searchsorted(xs[0], cx - r, side="left")
        xb = np.minimum(np.searchsorted(xs[0], cx + r, side="left") + 1, ow)
        ly, lx = int(max((yb - ya).max(), 0)), int(max((xb - xa).max(), 0))
        if ly > 0 and lx > 0:
            yi = ya[:, None] + np.arange(ly)[None, :]                    # (n, ly)
            xi = xa[:, None] + np.arange(lx)[None, :]                    # (n, lx)
            vy, vx = yi < yb[:, None], xi < xb[:, None]
            yc, xc = np.minimum(yi, oh - 1), np.minimum(xi, ow - 1)
            d2 = (ys[yc, 0] - cy[:, None])[:, :, None] ** 2 + (xs[0, xc] - cx[:, None])[:, None, :] ** 2
            hit = (d2 <= (r ** 2)[:, None, None]) & vy[:, :, None] & vx[:, None, :]
            kk, iy, ix = np.nonzero(hit)
            owner[yc[kk, iy], xc[kk, ix]] = kk
    return owner, cx, cy, r, ident, ys, xs


def render_region(seed: int, x0: int, y0: int, w: int, h: int) -> np.ndarray:
    """uint8 (h, w, 3) H&E-like pixels, pure function of absolute coordinates."""
    owner = _owner_map(seed, x0, y0, w, h)[0]
    base = np.where((owner >= 0)[..., None], FG, BG)
    ys = np.arange(y0, y0 + h, dtype=np.int64)[:, None, None] + (1 << 20)
    xs = np.arange(x0, x0 + w, dtype=np.int64)[None, :, None] + (1 << 20)
    cs = np.arange(3, dtype=np.int64)[None, None, :]
    u1 = _u01(_hash(seed, xs, ys, cs, 11))
    u2 = _u01(_hash(seed, xs, ys, cs, 12))
    noise = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2) * NOISE_SIGMA
    return np.clip(np.rint(base + noise), 0, 255).astype(np.uint8)


def render_points(seed: int, xs: np.ndarray, ys: np.ndarray) -> np.ndarray:
    """Pixels of the procedural slide at the grid xs x ys of absolute coordinates: uint8
    (len(ys), len(xs), 3), identical to ``render_region`` at those pixels."""
    xs, ys = np.asarray(xs, dtype=np.int64), np.asarray(ys, dtype=np.int64)
    out = np.empty((len(ys), len(xs), 3), np.uint8)
    gxs = np.floor_divide(xs, PITCH)
    gys = np.floor_divide(ys, PITCH)
    cs = np.arange(3, dtype=np.int64)[None, None, :]
    # nuclei of every grid cell the samples can touch, hashed once: [gy - gy0][gx - gx0]
    gx0, gx1, gy0, gy1 = int(gxs.min()) - 1, int(gxs.max()) + 1, int(gys.min()) - 1, int(gys.max()) + 1
    tgx, tgy = np.meshgrid(np.arange(gx0, gx1 + 1), np.arange(gy0, gy1 + 1), indexing="xy")
    kx, ky = tgx + (1 << 20), tgy + (1 << 20)
    tcx = (tgx + 0.5) * PITCH + (_u01(_hash(seed, kx, ky, 1)) * 2 - 1) * JITTER
    tcy = (tgy + 0.5) * PITCH + (_u01(_hash(seed, kx, ky, 2)) * 2 - 1) * JITTER
    tr2 = (R_MIN + _u01(_hash(seed, kx, ky, 3)) * (R_MAX - R_MIN)) ** 2
    rows = max(1, 500_000 // max(len(xs), 1))

    def band(r0):
        yb = ys[r0:r0 + rows]
        gyb = gys[r0:r0 + rows]
        inside = np.zeros((len(yb), len(xs)), bool)
        for dy in (-1, 0, 1):
            iy = (gyb + dy - gy0)[:, None]
            for dx in (-1, 0, 1):
                ix = (gxs + dx - gx0)[None, :]
                inside |= (yb[:, None] - tcy[iy, ix]) ** 2 + (xs[None, :] - tcx[iy, ix]) ** 2 <= tr2[iy, ix]
        base = np.where(inside[..., None], FG, BG)
        yk = yb[:, None, None] + (1 << 20)
        xk = xs[None, :, None] + (1 << 20)
        u1 = _u01(_hash(seed, xk, yk, cs, 11))
        u2 = _u01(_hash(seed, xk, yk, cs, 12))
        noise = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2) * NOISE_SIGMA
        out[r0:r0 + rows] = np.clip(np.rint(base + noise), 0, 255).astype(np.uint8)

    starts = list(range(0, len(ys), rows))
    if len(starts) > 1:                       # numpy releases the GIL: bands run on the host cores
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
            list(ex.map(band, starts))
    else:
        band(0)
    return out


def analytic_fields(seed: int, x0: int, y0: int, w: int, h: int, n_classes: int,
                    out_w: int | None = None, out_h: int | None = None):
    """Flow-injection tensors for the region, as the network would emit them.

    Returns dP (2,h,w) float32 [dY,dX] (x5 scale like the network output),
    cellprob (h,w) float32 (+6 inside, -6 outside), logits (n_classes,h,w)
    float32 (one-hot x4, class = id % (n_classes-1) + 1 inside, 0 outside) and
    the number of nuclei whose discs lie fully inside the region.  With out_w/out_h the
    fields are sampled on the region rescaled to that size (slide mpp != model mpp).
    """
    owner, cx, cy, r, ident, ys, xs = _owner_map(seed, x0, y0, w, h, out_w, out_h)
    rw, rh = w, h
    h, w = owner.shape
    inside = owner >= 0
    oc = np.where(inside, owner, 0)
    vy = np.where(inside, cy[oc] - ys, 0.0) if len(cx) else np.zeros((h, w))
    vx = np.where(inside, cx[oc] - xs, 0.0) if len(cx) else np.zeros((h, w))
    nrm = np.maximum(np.sqrt(vy * vy + vx * vx), 1.0)
    dP = np.stack((5.0 * vy / nrm, 5.0 * vx / nrm)).astype(np.float32)
    cellprob = np.where(inside, 6.0, -6.0).astype(np.float32)
    logits = np.zeros((n_classes, h, w), np.float32)
    if len(cx):
        cls = (ident % np.uint64(max(n_classes - 1, 1))).astype(np.int64) + 1
        cl_px = np.where(inside, cls[oc], 0)
    else:
        cl_px = np.zeros((h, w), np.int64)
    for c in range(n_classes):
        logits[c][cl_px == c] = 4.0
    full = (cx - r >= x0) & (cx + r <= x0 + rw - 1) & (cy - r >= y0) & (cy + r <= y0 + rh - 1)
    return dP, cellprob, logits, int(full.sum())


class SyntheticSlide:
    """OpenSlide-protocol reader over the procedural image (level 0 only)."""

    def __init__(self, width: int, height: int | None = None, mpp: float = 0.5,
                 seed: int = 1234, bounds: tuple[float, float] | None = None):
        self.width = int(width)
        self.height = int(height if height is not None else width)
        self.seed = int(seed)
        self.properties = {"openslide.mpp-x": str(mpp), "openslide.mpp-y": str(mpp)}
        if bounds is not None:
            self.properties["openslide.bounds-x"] = str(bounds[0])
            self.properties["openslide.bounds-y"] = str(bounds[1])
        self.level_count = 1
        self.level_dimensions = [(self.width, self.height)]
        self.level_downsamples = [1.0]
        self.dimensions = self.level_dimensions[0]

    @classmethod
    def from_uri(cls, uri: str) -> "SyntheticSlide":
        """``synthetic://<W>x<H>?mpp=0.5&seed=1234`` (or ``synthetic://<W>``)."""
        body = uri.split("://", 1)[1]
        dims, _, query = body.partition("?")
        dims = dims.rstrip("/").split(".")[0]
        w, _, h = dims.partition("x")
        kw = dict(p.split("=", 1) for p in query.split("&") if "=" in p)
        return cls(int(w), int(h) if h else None, float(kw.get("mpp", 0.5)),
                   int(kw.get("seed", 1234)))

    def get_best_level_for_downsample(self, downsample: float) -> int:
        return 0

    def read_region(self, location, level: int, size) -> np.ndarray:
        x0, y0 = int(location[0]), int(location[1])
        w, h = int(size[0]), int(size[1])
        rgb = render_region(self.seed, x0, y0, w, h)
        return np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=-1)

    def get_thumbnail(self, size) -> np.ndarray:
        """nearest-pixel subsample of level 0 (the pixel function is evaluated at the sampled
        coordinates only, so 10^4-px thumbnails of 10^4..10^5-px slides stay cheap)"""
        w, h = int(size[0]), int(size[1])
        xs = (np.arange(w) * (self.width / w)).astype(np.int64)
        ys = (np.arange(h) * (self.height / h)).astype(np.int64)
        return render_points(self.seed, xs, ys)

    def close(self) -> None:
        pass


# --------------------------------------------------------------------------
# seeded weights with the reference's key layout
# --------------------------------------------------------------------------
def make_state_dict(n_classes: int = 7, fts: list[int] | None = None, depth: int = 24,
                    seed: int = 0, embed: int = 1024, mlp_ratio: int = 4):
    """Random-init ClassTransformer state dict (float32 torch tensors).

    Scales are chosen so activations stay O(1) through ``depth`` blocks (LN'd
    residual stream, small residual branches), which keeps bf16-vs-fp32
    comparisons meaningful.  rel_pos table heights follow SAM ViT-L: 127 rows
    in the originally-global blocks (5, 11, 17, 23), 27 elsewhere.
    """
    import torch
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    hd = 64
    sd = {}
    sd["encoder.patch_embed.proj.weight"] = rn(embed, 3, 8, 8, std=1.0 / np.sqrt(192))
    sd["encoder.patch_embed.proj.bias"] = rn(embed, std=0.02)
    sd["encoder.pos_embed"] = rn(1, 32, 32, embed, std=0.2)
    for i in range(depth):
        p = f"encoder.blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + rn(embed, std=0.05)
        sd[p + "norm1.bias"] = rn(embed, std=0.02)
        sd[p + "attn.qkv.weight"] = rn(3 * embed, embed, std=1.0 / np.sqrt(embed))
        sd[p + "attn.qkv.bias"] = rn(3 * embed, std=0.02)
        sd[p + "attn.proj.weight"] = rn(embed, embed, std=0.5 / np.sqrt(embed))
        sd[p + "attn.proj.bias"] = rn(embed, std=0.02)
        nrel = 127 if i in (5, 11, 17, 23) else 27
        sd[p + "attn.rel_pos_h"] = rn(nrel, hd, std=0.15)
        sd[p + "attn.rel_pos_w"] = rn(nrel, hd, std=0.15)
        sd[p + "norm2.weight"] = 1.0 + rn(embed, std=0.05)
        sd[p + "norm2.bias"] = rn(embed, std=0.02)
        sd[p + "mlp.lin1.weight"] = rn(mlp_ratio * embed, embed, std=1.0 / np.sqrt(embed))
        sd[p + "mlp.lin1.bias"] = rn(mlp_ratio * embed, std=0.02)
        sd[p + "mlp.lin2.weight"] = rn(embed, mlp_ratio * embed, std=0.5 / np.sqrt(mlp_ratio * embed))
        sd[p + "mlp.lin2.bias"] = rn(embed, std=0.02)
    sd["encoder.neck.0.weight"] = rn(256, embed, 1, 1, std=1.0 / np.sqrt(embed))
    sd["encoder.neck.1.weight"] = 1.0 + rn(256, std=0.05)
    sd["encoder.neck.1.bias"] = rn(256, std=0.02)
    sd["encoder.neck.2.weight"] = rn(256, 256, 3, 3, std=1.0 / np.sqrt(256 * 9))
    sd["encoder.neck.3.weight"] = 1.0 + rn(256, std=0.05)
    sd["encoder.neck.3.bias"] = rn(256, std=0.02)
    sd["out.weight"] = rn(192, 256, 1, 1, std=1.0 / np.sqrt(256))
    sd["out.bias"] = rn(192, std=0.1)
    sd["W2"] = torch.eye(192).reshape(192, 3, 8, 8)
    sd["diam_labels"] = torch.tensor([30.0])
    sd["diam_mean"] = torch.tensor([30.0])
    if n_classes <= 1:                       # plain Cellpose-SAM checkpoint (cpsam): no semantic head
        return sd
    oc = n_classes * 64
    if fts is None:
        sd["out_class.weight"] = rn(oc, 256, 1, 1, std=1.0 / np.sqrt(256))
        sd["out_class.bias"] = rn(oc, std=0.1)
    else:
        def conv(pfx, cin, cout, k):
            sd[pfx + ".weight"] = rn(cout, cin, k, k, std=1.0 / np.sqrt(cin * k * k))
            sd[pfx + ".bias"] = rn(cout, std=0.05)

        def convT(pfx, cin, cout, k):
            sd[pfx + ".weight"] = rn(cin, cout, k, k, std=1.0 / np.sqrt(cin))
            sd[pfx + ".bias"] = rn(cout, std=0.05)

        ins = [256, *fts]
        outs = [*fts[::-1], oc]
        for n, (ci, co) in enumerate(zip(ins[:-1], ins[1:])):
            p = f"out_class.encoder_blocks.{n}."
            conv(p + "block.conv1", ci, co, 3)
            conv(p + "block.conv2", co, co, 3)
            conv(p + "downconv", co, co, 2)
        for n, (ci, co) in enumerate(zip(outs[:-1], outs[1:])):
            p = f"out_class.decoder_blocks.{n}."
            conv(p + "block.conv1", 2 * ci, co, 3)
            conv(p + "block.conv2", co, co, 3)
            convT(p + "upconv", co, co, 2)
        c = ins[-1]
        conv("out_class.bottleneck_down.block.conv1", c, c, 3)
        conv("out_class.bottleneck_down.block.conv2", c, c, 3)
        conv("out_class.bottleneck_down.downconv", c, c, 2)
        conv("out_class.bottleneck_up.block.conv1", c, c, 3)
        conv("out_class.bottleneck_up.block.conv2", c, c, 3)
        convT("out_class.bottleneck_up.upconv", c, c, 2)
    sd["W3"] = torch.eye(oc).reshape(oc, n_classes, 8, 8)
    return sd


def make_grandqc_state_dict(n_classes: int = 2, seed: int = 0):
    """Seeded random UNet++/EfficientNet-B0 state dict with the key layout smp/timm produce
    (``classpose_amd.qc_arch.expected_shapes``): He-style conv scales, BatchNorm statistics near
    identity, so activations stay O(1) through the ~100 layers."""
    import torch
    from . import qc_arch
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in qc_arch.expected_shapes(n_classes).items():
        leaf = k.rsplit(".", 1)[1]
        if leaf == "running_var":
            sd[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            sd[k] = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1 and leaf == "weight":                 # BN gamma
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                                      # BN beta / conv bias
            sd[k] = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[k] = torch.randn(shape, generator=g) * (1.6 / fan_in) ** 0.5
    # tensors real checkpoints carry but the forward pass never reads
    sd["encoder.conv_head.weight"] = torch.zeros(1280, 320, 1, 1)
    for n in ("weight", "bias", "running_mean", "running_var"):
        sd["encoder.bn2." + n] = torch.ones(1280)
    return sd


def analytic_qc_map(kind: str, height: int, width: int) -> np.ndarray:
    """Class maps for QC-injection tests on synthetic slides, in thumbnail pixels.

    tissue: class 0 (tissue) inside an ellipse covering the central part of the slide with a
    circular hole, class 1 (background) elsewhere.  artefact: class 1 (normal tissue) everywhere
    except a rectangle of class 2 (fold) in the upper-left part of the ellipse."""
    yy, xx = np.mgrid[0:height, 0:width]
    u, v = (xx + 0.5) / width, (yy + 0.5) / height
    ell = ((u - 0.5) / 0.40) ** 2 + ((v - 0.5) / 0.36) ** 2 <= 1.0
    hole = (u - 0.62) ** 2 + ((v - 0.55) * height / width) ** 2 <= 0.06 ** 2
    if kind == "tissue":
        return np.where(ell & ~hole, 0, 1).astype(np.int8)
    fold = (u > 0.30) & (u < 0.42) & (v > 0.30) & (v < 0.45)
    return np.where(fold, 2, 1).astype(np.int8)


def register(hooks, argument: str) -> None:
    """Plug-in entry (classpose_amd/hooks.py): ``CLASSPOSE_AMD_PLUGINS=classpose_amd.synth:flow``, ``:qc`` or
    ``:flow+qc``.  flow: the dynamics of tiles read from a SyntheticSlide consume the analytic fields of its procedural
    nuclei (other readers are left alone); qc: the GrandQC class maps come from ``analytic_qc_map``."""
    what = set(filter(None, (argument or "flow").split("+")))
    unknown = what - {"flow", "qc"}
    if unknown:
        raise ValueError(f"classpose_amd.synth plug-in: unknown option(s) {sorted(unknown)}")
    if "flow" in what:
        def field_provider(slide, plan, n_classes):
            if not hasattr(slide, "seed"):
                return None
            return lambda ti, R, W, H: analytic_fields(slide.seed, plan.coords[ti][0][0], plan.coords[ti][0][1], R, R,
                                                       n_classes, W, H)
        hooks.field_provider = field_provider
    if "qc" in what:
        hooks.qc_provider = lambda kind: (lambda image: analytic_qc_map(kind, image.shape[0], image.shape[1]))
