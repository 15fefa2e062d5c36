"""Device pipeline for the WSI tile path: uint8 tiles in HBM -> instance/class maps.

Host-side mirror of what ``ClassposeModel.eval`` does per tile
(/root/reference/src/classpose/models.py:478-827: normalize -> core.run_net ->
compute_masks -> compute_class_masks), re-cut for MI355X: tiles are batched
ACROSS WSI tiles (the reference only batches the sub-tiles of one tile,
predict_wsi.py:749-756), every stage is a hand-written HIP kernel behind the C
ABI of include/classpose_hip.h, and nothing returns to the host between stages.
torch is used for device memory, streams and (elsewhere) torch.distributed only.
"""
from __future__ import annotations

import os

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import CpxBlockWeights, CpxConvOp, CpxNetWeights, CpxRecord, CpxTiling, check, ptr

PS = 8
BSIZE = 256
HALF_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}
NET_DTYPES = {**HALF_DTYPES, "fp32": torch.float32}      # resolve_precision, models.py:37-69


# --------------------------------------------------------------------------
# host geometry (cellpose get_pad_yx / make_tiles grid, core.py:129-149)
# --------------------------------------------------------------------------
def make_tiling(H: int, W: int, bsize: int = BSIZE, augment: bool = False,
                tile_overlap: float = 0.1) -> CpxTiling:
    def pad(L):
        lpad = int(16 * math.ceil(L / 16) - L) if L >= bsize else bsize - L
        return 8 + lpad // 2, 8 + lpad - lpad // 2
    yp1, yp2 = pad(H)
    xp1, xp2 = pad(W)
    Ly, Lx = H + yp1 + yp2, W + xp1 + xp2
    if augment:
        ny = max(2, int(math.ceil(2.0 * Ly / bsize)))
        nx = max(2, int(math.ceil(2.0 * Lx / bsize)))
    else:
        ov = min(0.5, max(0.05, tile_overlap))
        ny = 1 if Ly <= bsize else int(math.ceil((1.0 + 2 * ov) * Ly / bsize))
        nx = 1 if Lx <= bsize else int(math.ceil((1.0 + 2 * ov) * Lx / bsize))
    if ny > 16 or nx > 16:
        raise ValueError(f"tile {H}x{W} needs a {ny}x{nx} sub-tile grid; at most 16x16 is supported")
    t = CpxTiling()
    t.H, t.W, t.ypad1, t.xpad1, t.Ly, t.Lx = H, W, yp1, xp1, Ly, Lx
    t.ny, t.nx, t.bsize, t.augment = ny, nx, bsize, int(bool(augment))
    ys = np.linspace(0, Ly - bsize, ny).astype(int)
    xs = np.linspace(0, Lx - bsize, nx).astype(int)
    for i in range(16):
        t.ystart[i] = int(ys[i]) if i < ny else 0
        t.xstart[i] = int(xs[i]) if i < nx else 0
    return t


def taper_1d(bsize: int = BSIZE, sig: float = 7.5) -> np.ndarray:
    """1-D factor of cellpose's ``_taper_mask`` for a bsize x bsize sub-tile (float64)."""
    b = max(224, bsize)
    xm = np.arange(b)
    xm = np.abs(xm - xm.mean())
    m = 1 / (1 + np.exp((xm - (b / 2 - 20)) / sig))
    return np.ascontiguousarray(m[b // 2 - bsize // 2: b // 2 + bsize // 2 + bsize % 2])


def percentile_params(n: int, q: float) -> tuple[int, float]:
    """(previous index, gamma) of np.percentile(..., q) on n float32 samples:
    numpy's 'linear' method evaluated in float32 exactly like numpy does."""
    q32 = np.float32(q) / np.float32(100)
    vi = np.float32((n - 1) * q32)      # _QuantileMethods["linear"]: (n - 1) * quantiles, in float32
    prev = np.floor(vi)
    if vi >= n - 1:
        return n - 1, 0.0
    return int(prev), float(np.float32(vi - prev))


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def interp_rel_pos(rel_pos: torch.Tensor, size: int = 63) -> torch.Tensor:
    """segment_anything get_rel_pos resize step (F.interpolate linear) in rel_pos.dtype."""
    if rel_pos.shape[0] == size:
        return rel_pos
    r = torch.nn.functional.interpolate(
        rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=size, mode="linear")
    return r.reshape(-1, size).permute(1, 0)


class NetWeights:
    """ClassTransformer parameters laid out for the HIP kernels.

    ``from_state_dict`` accepts the reference's state-dict layout
    (predict_wsi.py:1393-1405; ``torch.save(net.state_dict())``, vit_sam.py:269-285)
    and mirrors ``net.load_model`` + ``net.to(dtype)``: every parameter is first
    rounded to the network dtype, GEMM weights stay in it, vectors are widened back
    to float32 for the epilogues.  ``fp32`` keeps everything float32 (exact-f32 MFMA
    kernels, csrc/cpx_net_f32.hip) and never folds LayerNorm.
    """

    def __init__(self):
        self.keep = []          # tensors that own the device memory
        self.c = CpxNetWeights()

    @staticmethod
    def infer_structure(sd) -> tuple[list[int] | None, int, int]:
        import re
        fts = [sd[k].shape[0] for k in sd
               if re.search(r"out_class\.encoder_blocks\.[0-9]+\.block.conv1.weight", k)]
        n_classes = sd["W3"].shape[1] if "W3" in sd else 1
        depth = 1 + max(int(m.group(1)) for k in sd
                        if (m := re.match(r"encoder\.blocks\.(\d+)\.norm1\.weight", k)))
        return (fts or None), n_classes, depth

    @classmethod
    def from_state_dict(cls, sd: dict, precision: str = "bf16", device="cuda:0",
                        fuse_ln: bool = True) -> "NetWeights":
        if precision not in NET_DTYPES:
            raise ValueError(f"precision {precision!r}: expected one of {sorted(NET_DTYPES)}")
        sd = {k.removeprefix("module."): v for k, v in sd.items()}
        hd = NET_DTYPES[precision]
        if precision == "fp32":
            fuse_ln = False
        fts, ncls, depth = cls.infer_structure(sd)
        if fts is not None and len(fts) > 4:
            # 32 x 32 tokens halve once per encoder level and once more in the bottleneck: a fifth level would hand
            # torch's own Conv2d(k=2, s=2) a 0 x 0 image in the reference too (unet.py:173-196)
            raise ValueError("UNet semantic head: at most 4 encoder levels fit the 32 x 32 token grid")
        if sd["encoder.patch_embed.proj.weight"].shape != (1024, 3, 8, 8):
            raise ValueError("only the vit_l / ps=8 Cellpose-SAM backbone is supported")
        self = cls()
        dev = torch.device(device)

        # Round 6: the checkpoint is uploaded as stored (float32) and rounded / folded ON THE DEVICE (csrc/cpx_weights.hip).  Until
        # round 5 the host did both: 0.6 s at one rank, 3.5 - 4.7 s per rank with eight ranks on a 16-core quota.  The helpers below
        # only LIST what every parameter becomes (output tensors allocated, one cpx_weight_job each); ONE foreign call at the end
        # (cpx_weights_build) streams the sources host -> staging and queues the kernels -- per-tensor calls re-acquired the
        # interpreter lock ~600 times against the slide readers' threads: 2.5 s per rank at eight ranks for 0.3 s of work.
        L = _lib.lib()
        dtc = _lib.DTYPE_CODE[precision]
        on_gpu = dev.type == "cuda"
        if not on_gpu and precision != "fp32":
            # (float32 on a CPU device is the host-side packing only -- tests/test_host_logic.py reads the op list back; nothing rounds)
            raise _lib.CpxError("NetWeights: the half-precision operands are rounded and folded by HIP kernels; there is no CPU path")
        jobs, sources, stage_bytes = [], [], 0

        def host32(t):      # contiguous float32 host view of a parameter (a bf16 / fp16 checkpoint widens exactly), kept alive until the call
            x = t.detach()
            x = (x if x.dtype == torch.float32 else x.float()).contiguous()
            sources.append(x)
            return x

        def job(op, srcs, dsts, n, K=0):
            nonlocal stage_bytes
            j = _lib.CpxWeightJob()
            j.op, j.dtype, j.n, j.K = op, dtc, n, K
            for k, x in enumerate(srcs):
                j.src_host[k] = x.data_ptr()
                if op != _lib.WJ_COPY_F32:
                    j.stage_off[k] = stage_bytes
                    stage_bytes += (x.numel() * 4 + 255) // 256 * 256
            for k, d in enumerate(dsts):
                j.dst[k] = d.data_ptr()
            jobs.append(j)
            self.keep += dsts

        def on_device(t, keep_f32, exact=False):
            x = host32(t)
            if not on_gpu:
                self.keep.append(x)
                return x.data_ptr()
            as_is = exact or hd == torch.float32
            out = torch.empty(x.shape, dtype=torch.float32 if (keep_f32 or as_is) else hd, device=dev)
            if x.numel():
                job(_lib.WJ_COPY_F32 if as_is else _lib.WJ_ROUND_F32 if keep_f32 else _lib.WJ_ROUND_HALF, [x], [out], x.numel())
            else:
                self.keep.append(out)
            return out.data_ptr()

        def half(t):        # GEMM operand: stays in the half dtype
            return on_device(t, False)

        def vec(t):         # epilogue vector: rounded through the half dtype, kept as f32
            return on_device(t, True)

        c = self.c
        c.depth, c.ncls = depth, ncls
        c.n_head_cols = 192 + (ncls * 64 if ncls > 1 else 0)
        c.ld_head = (c.n_head_cols + 127) // 128 * 128
        c.dtype = _lib.DTYPE_CODE[precision]
        c.prof = None
        c.fuse_ln = int(bool(fuse_ln))

        def fold_ln(w, b, gamma, beta):
            """LayerNorm folded into the following Linear: (W diag(gamma), b + W beta, row sums of the folded half-rounded W) as
            device pointers.  All inputs are first rounded to the half dtype (net.to(dtype)); cpx_fold_layernorm, float64 sums."""
            N, K = w.shape
            wf = torch.empty((N, K), dtype=hd, device=dev)
            bf = torch.empty(N, dtype=torch.float32, device=dev)
            cs = torch.empty(N, dtype=torch.float32, device=dev)
            job(_lib.WJ_FOLD_LN, [host32(t) for t in (w, b, gamma, beta)], [wf, bf, cs], N, K)
            return wf.data_ptr(), bf.data_ptr(), cs.data_ptr()

        def vec32(t):       # already float32, no re-rounding
            return on_device(t, True, exact=True)
        c.pe_w = half(sd["encoder.patch_embed.proj.weight"].reshape(1024, 192))
        c.pe_b = vec(sd["encoder.patch_embed.proj.bias"])
        c.pos = vec(sd["encoder.pos_embed"].reshape(1024, 1024))
        self.blocks = (CpxBlockWeights * depth)()
        for i in range(depth):
            p = f"encoder.blocks.{i}."
            b = self.blocks[i]
            b.ln1_w, b.ln1_b = vec(sd[p + "norm1.weight"]), vec(sd[p + "norm1.bias"])
            if fuse_ln:
                wf, bf, cs = fold_ln(sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
                                     sd[p + "norm1.weight"], sd[p + "norm1.bias"])
                b.qkv_w, b.qkv_b, b.qkv_colsum = wf, bf, cs
            else:
                b.qkv_w, b.qkv_b = half(sd[p + "attn.qkv.weight"]), vec(sd[p + "attn.qkv.bias"])
            b.proj_w, b.proj_b = half(sd[p + "attn.proj.weight"]), vec(sd[p + "attn.proj.bias"])
            for name, key in (("rel_h", "attn.rel_pos_h"), ("rel_w", "attn.rel_pos_w")):
                # table interpolated to 2*32-1 rows in the half dtype (what get_rel_pos does on
                # the casted parameter), x8 = 1/scale (exact), padded with a zero row to 64
                t = interp_rel_pos(sd[p + key].detach().to(hd)).float() * 8.0
                t = torch.cat([t, torch.zeros(1, 64)], 0)
                setattr(b, name, half(t))
            b.ln2_w, b.ln2_b = vec(sd[p + "norm2.weight"]), vec(sd[p + "norm2.bias"])
            if fuse_ln:
                wf, bf, cs = fold_ln(sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"],
                                     sd[p + "norm2.weight"], sd[p + "norm2.bias"])
                b.fc1_w, b.fc1_b, b.fc1_colsum = wf, bf, cs
            else:
                b.fc1_w, b.fc1_b = half(sd[p + "mlp.lin1.weight"]), vec(sd[p + "mlp.lin1.bias"])
            b.fc2_w, b.fc2_b = half(sd[p + "mlp.lin2.weight"]), vec(sd[p + "mlp.lin2.bias"])
        c.blocks = C.cast(self.blocks, C.POINTER(CpxBlockWeights))
        c.neck0_w = half(sd["encoder.neck.0.weight"].reshape(256, 1024))
        c.neck_ln1_w, c.neck_ln1_b = vec(sd["encoder.neck.1.weight"]), vec(sd["encoder.neck.1.bias"])
        c.neck2_w = half(sd["encoder.neck.2.weight"].permute(0, 2, 3, 1).reshape(256, 2304))
        c.neck_ln2_w, c.neck_ln2_b = vec(sd["encoder.neck.3.weight"]), vec(sd["encoder.neck.3.bias"])
        hw = [sd["out.weight"].reshape(192, 256)]
        hb = [sd["out.bias"]]
        if fts is not None:
            hw.append(torch.zeros(ncls * 64, 256))       # class columns come from the UNet head below
            hb.append(torch.zeros(ncls * 64))
        elif ncls > 1:
            hw.append(sd["out_class.weight"].reshape(ncls * 64, 256))
            hb.append(sd["out_class.bias"])
        hw = torch.cat(hw, 0)
        hb = torch.cat(hb, 0)
        padn = c.ld_head - hw.shape[0]
        c.head_w = half(torch.cat([hw, torch.zeros(padn, 256)], 0))
        c.head_b = vec(torch.cat([hb, torch.zeros(padn)], 0))
        c.n_unet_ops = 0
        if fts is not None:
            self._build_unet_ops(sd, fts, ncls * 64, half, vec32)
        self.ncls, self.depth, self.precision, self.device = ncls, depth, precision, dev
        self.fts = fts
        if jobs:
            with torch.cuda.device(dev):                  # (the kernels are launched on this thread's CURRENT device: it must be the one the stream belongs to)
                stage = torch.empty(stage_bytes, dtype=torch.uint8, device=dev)
                arr = (_lib.CpxWeightJob * len(jobs))(*jobs)
                st = torch.cuda.current_stream(dev)
                check(L.cpx_weights_build(arr, len(jobs), stage.data_ptr(), stage_bytes, st.cuda_stream), "weights_build")
                st.synchronize()                          # the operands are final before any other stream may read them; sources + staging may go
                del stage
        sources.clear()
        return self

    def _build_unet_ops(self, sd, fts, out_ch, half, vec32):
        """Flatten classpose.unet.UNet (unet.py:121-196, any ``n_channels`` list) into the conv list of cpx_conv_op.

        The device kernels move 16-byte channel chunks, so every tensor carries its channel count rounded up to a
        multiple of 8: the extra input columns and output rows of each weight matrix (and the extra biases) are zero,
        hence the extra channels hold exact zeros through every ReLU / concat and contribute nothing downstream."""
        hd = {0: torch.bfloat16, 1: torch.float16, 2: torch.float32}[self.c.dtype]
        up = lambda x, m: (x + m - 1) // m * m
        p8 = lambda c: up(c, 8)
        ops = []

        def rounded(t):
            return t.detach().to(hd).float()

        def add(kind, src_a, src_b, cin_a, cin_b, cout, h, relu, wkey):
            w, b = rounded(sd[wkey + ".weight"]), rounded(sd[wkey + ".bias"])
            ca, cb, co = p8(cin_a), p8(cin_b), p8(cout)
            if kind == 2:       # ConvTranspose2d [cin][cout][2][2] -> rows (dy, dx, co), cols ci
                wt = torch.zeros(2, 2, co, ca)
                wt[:, :, :cout, :cin_a] = w.permute(2, 3, 1, 0)
                wm = wt.reshape(4 * co, ca)
                bt = torch.zeros(4, co)
                bt[:, :cout] = b[None, :]
                bm = bt.reshape(-1)
            else:               # Conv2d [cout][cin_a + cin_b][k][k] -> cols (ky, kx, ci of a | ci of b)
                k = w.shape[-1]
                wt = torch.zeros(co, k, k, ca + cb)
                wk = w.permute(0, 2, 3, 1)
                wt[:cout, :, :, :cin_a] = wk[..., :cin_a]
                wt[:cout, :, :, ca:ca + cin_b] = wk[..., cin_a:]
                wm = wt.reshape(co, -1)
                bm = torch.zeros(co)
                bm[:cout] = b
            n_pad, k_pad = up(wm.shape[0], 128), up(wm.shape[1], 64)
            wp = torch.zeros(n_pad, k_pad)
            wp[: wm.shape[0], : wm.shape[1]] = wm
            bp = torch.zeros(n_pad)
            bp[: bm.shape[0]] = bm
            op = CpxConvOp()
            op.kind, op.src_a, op.src_b, op.dst = kind, src_a, src_b, len(ops) + 1
            op.cin_a, op.cin_b, op.cout, op.h, op.w, op.relu = ca, cb, co, h, h, int(relu)
            op.weight, op.bias = half(wp), vec32(bp)
            ops.append(op)
            return op.dst

        def block(pfx, src_a, src_b, cin_a, cin_b, cout, h, last_relu=True):
            t = add(0, src_a, src_b, cin_a, cin_b, cout, h, True, pfx + "block.conv1")
            return add(0, t, -1, cout, 0, cout, h, last_relu, pfx + "block.conv2")

        cur, cin, h, feats = 0, 256, 32, []
        for n, c in enumerate(fts):
            p = f"out_class.encoder_blocks.{n}."
            t = block(p, cur, -1, cin, 0, c, h)
            cur = add(1, t, -1, c, 0, c, h, False, p + "downconv")
            h //= 2
            feats.append((cur, c))
            cin = c
        c = fts[-1]
        t = block("out_class.bottleneck_down.", cur, -1, c, 0, c, h)
        cur = add(1, t, -1, c, 0, c, h, False, "out_class.bottleneck_down.downconv")
        h //= 2
        t = block("out_class.bottleneck_up.", cur, -1, c, 0, c, h)
        cur = add(2, t, -1, c, 0, c, h, False, "out_class.bottleneck_up.upconv")
        h *= 2
        seq = [*fts[::-1], out_ch]
        feats = feats[::-1]
        for i in range(len(fts)):
            p = f"out_class.decoder_blocks.{i}."
            fid, fc = feats[i]
            t = block(p, cur, fid, seq[i], fc, seq[i + 1], h, last_relu=(i != len(fts) - 1))
            cur = add(2, t, -1, seq[i + 1], 0, seq[i + 1], h, False, p + "upconv")
            h *= 2
        if h != 32:
            raise ValueError(f"UNet head: {len(fts)} levels do not end on the 32 x 32 token grid (reached {h})")
        if out_ch % 8 != 0:
            # the last layer writes 16-byte chunks of out_ch columns (8 x 8 pixel shuffle per class): a checkpoint whose
            # class count breaks that must be rejected here, not reach the device kernels unpadded (asserts vanish under -O)
            raise ValueError(f"UNet head: out_ch = {out_ch} is not a multiple of 8")
        self.unet_ops = (CpxConvOp * len(ops))(*ops)
        self.c.n_unet_ops = len(ops)
        self.c.unet_ops = C.cast(self.unet_ops, C.POINTER(CpxConvOp))


# --------------------------------------------------------------------------
# engine
# --------------------------------------------------------------------------
@dataclass
class TileOutputs:
    masks: torch.Tensor          # uint16 [nT, H, W]   (torch has no uint16 math; stored as int16 bits)
    class_masks: torch.Tensor    # uint8  [nT, H, W]
    nlabels: torch.Tensor        # int32  [nT]
    dP: torch.Tensor             # f32 [nT, 2, H, W]
    cellprob: torch.Tensor       # f32 [nT, H, W]
    logits: torch.Tensor | None  # f32 [nT, ncls, H, W]
    records: torch.Tensor | None = None      # uint8 view of cpx_record[nT][max_rec]
    rec_counts: torch.Tensor | None = None   # int32 [nT]
    cells: torch.Tensor | None = None        # uint8 view of cpx_cell[nT][max_rec]   (polygons=...)
    xy_pool: torch.Tensor | None = None      # f64 [max_pts, 2] level-0 vertices
    n_pts_total: torch.Tensor | None = None  # int32 [1]


class _Slot:
    """One in-flight batch: every buffer a batch touches after the shared network workspace."""

    def __init__(self, eng: "Engine"):
        nT, H, W, ncls, d = eng.nT, eng.H, eng.W, eng.w.ncls, eng.dev
        nS = nT * eng.n_sub
        self.stats = torch.empty(nT * 3 * 4, dtype=torch.float32, device=d)
        self.hist = torch.empty(nT * 768, dtype=torch.int32, device=d)
        # zeros: a partial first batch still runs all rows
        self.patches = torch.zeros(nS * 1024 * 192 * (2 if eng.w.c.dtype == _lib.DT_F32 else 1), dtype=torch.int16, device=d)
        self.head = torch.empty(nS * 1024 * eng.w.c.ld_head, dtype=torch.float32, device=d)
        self.dP = torch.empty((nT, 2, H, W), dtype=torch.float32, device=d)
        self.cellprob = torch.empty((nT, H, W), dtype=torch.float32, device=d)
        self.logits = torch.empty((nT, max(ncls, 1), H, W), dtype=torch.float32, device=d)
        self.pp_ws = torch.empty(eng.pp_ws_bytes, dtype=torch.uint8, device=d)
        self.masks = torch.empty((nT, H, W), dtype=torch.int16, device=d)
        self.class_masks = torch.empty((nT, H, W), dtype=torch.uint8, device=d)
        self.nlabels = torch.empty(nT, dtype=torch.int32, device=d)
        self.records = torch.empty(nT * eng.max_rec * C.sizeof(CpxRecord), dtype=torch.uint8, device=d)
        self.rec_counts = torch.empty(nT, dtype=torch.int32, device=d)
        self.cells = self.xy_pool = self.n_pts_total = self.poly_ws = self.origins = None   # allocated on first use
        self.has_polygons = False
        self.ev_net = torch.cuda.Event()
        self.ev_post = torch.cuda.Event()
        self.n = 0
        self.inject = None
        self.busy = False


class Engine:
    """Persistent device buffers + launch sequence for batches of nT WSI tiles.

    Two HIP streams: the NETWORK stream runs normalise -> sub-tile -> ClassTransformer of batch
    i+1 while the POST stream runs blend -> dynamics -> class vote -> records of batch i (the
    post-processing kernels are small and latency-bound: they slot into the gaps of the MFMA
    kernels).  ``submit`` / ``result`` expose the pipeline; ``run`` = submit + result.
    """

    N_SLOTS = 2

    def __init__(self, weights: NetWeights, tile_h: int = 256, tile_w: int | None = None,
                 batch_tiles: int = 8, augment: bool = False, tile_overlap: float = 0.1,
                 niter: int = 200, cellprob_threshold: float = 0.0, flow_threshold: float = 0.4,
                 min_size: int = 15, max_size_fraction: float = 0.4):
        self.L = _lib.lib()
        self.w = weights
        self.dev = weights.device
        self.H, self.W = tile_h, tile_w or tile_h
        self.nT = batch_tiles
        self.tiling = make_tiling(self.H, self.W, BSIZE, augment, tile_overlap)
        self.n_sub = self.tiling.ny * self.tiling.nx
        self.niter, self.cp_thr, self.flow_thr = niter, cellprob_threshold, flow_threshold
        self.min_size, self.max_frac = min_size, max_size_fraction
        nT, H, W = self.nT, self.H, self.W
        d = self.dev
        lo = percentile_params(H * W, 1)
        hi = percentile_params(H * W, 99)
        self.pct = (lo[0], lo[1], hi[0], hi[1])
        self.net_ws_bytes = self.L.cpx_net_workspace_bytes(nT * self.n_sub, weights.c.dtype)
        if weights.c.n_unet_ops:
            self.net_ws_bytes += self.L.cpx_unet_workspace_bytes(weights.c.unet_ops, weights.c.n_unet_ops,
                                                                nT * self.n_sub, weights.c.dtype)
        self.net_ws = torch.empty(self.net_ws_bytes, dtype=torch.uint8, device=d)
        self.taper = torch.from_numpy(taper_1d(BSIZE)).to(d)
        self.pp_ws_bytes = self.L.cpx_postproc_workspace_bytes(nT, H, W)
        # every label the dynamics can produce gets a record slot; ids are uint16, so 65535 bounds it
        # (fetch_* raise instead of dropping cells when a tile would exceed the buffer)
        self.max_rec = min(self.L.cpx_postproc_max_labels(H, W), 65535)
        # device vertex pool (f1), 16 B per vertex.  A contour visits a pixel at most twice (a one-pixel-wide limb is walked up one side and down the
        # other), so 2 H W per tile can never overflow: 16.8 MB per slot for 8 tiles of 256 px.  (H W / 8 until round 5: a tile of 500 - 800 small
        # nuclei produces 16 - 20 vertices each and overflowed it -- fetch_polygons then hands the whole batch to the host polygoniser,
        # tools/poly_size_scan.py.)  Only the vertices a batch produced are copied back.
        self.max_pts = nT * max(4096, 2 * H * W)
        self.slots = [_Slot(self) for _ in range(self.N_SLOTS)]
        self.s_net = torch.cuda.Stream(d, priority=int(os.environ.get("CPX_NET_STREAM_PRIORITY", "0")))
        # the post-processing chain is ~38 short kernels that run beside persistent network kernels holding every CU: a
        # higher stream priority lets their workgroups take a CU the moment one frees instead of queueing behind the
        # network's next tiles (CPX_POST_STREAM_PRIORITY: -1 high, 0 normal; A/B in tools/ab_post_priority.py)
        self.s_post = torch.cuda.Stream(d, priority=int(os.environ.get("CPX_POST_STREAM_PRIORITY", "0")))
        self._next = 0
        self._last: _Slot | None = None
        # None, or a list that submit() appends five timing events per batch to: [pre start, pre end = network start, network end] on the
        # network stream, [blend start, blend end] on the post stream (bench.py: roofline.other_kernels_ms_per_step)
        self.stage_timing: list | None = None

    # -- pipeline -------------------------------------------------------
    def submit(self, tiles_u8: torch.Tensor, inject=None, records: bool = True, polygons=None) -> int:
        """Enqueue one batch (uint8 [n, H, W, 3] resident on the device, n <= batch_tiles) and
        return its slot id.  ``inject`` = (dP, cellprob, logits) device tensors: flow-injection
        mode (the network still runs; the dynamics consume the injected fields instead).
        ``polygons`` = (scale, origins [n][2] level-0 tile origins): also polygonise the instances
        on the device (``cpx_polygonize_device``), results via ``fetch_polygons``."""
        # (ValueError, not assert: these guard raw device pointers handed to the C ABI and must survive `python -O`)
        if tiles_u8.dtype != torch.uint8 or tiles_u8.dim() != 4 or not tiles_u8.is_contiguous():
            raise ValueError(f"Engine.submit: tiles must be a contiguous uint8 [n, H, W, 3] tensor, got {tiles_u8.dtype} {tuple(tiles_u8.shape)}")
        n = tiles_u8.shape[0]
        if n > self.nT or tuple(tiles_u8.shape[1:]) != (self.H, self.W, 3):
            raise ValueError(f"Engine.submit: got {tuple(tiles_u8.shape)}, this engine takes at most {self.nT} tiles of {(self.H, self.W, 3)}")
        if tiles_u8.device != self.dev:
            raise ValueError(f"Engine.submit: tiles live on {tiles_u8.device}, the engine on {self.dev}")
        if inject is not None:
            # logits may be None for a model without a class head (ncls <= 1: the chain never takes its pointer); dP and cellprob may not
            if len(inject) != 3 or inject[0] is None or inject[1] is None or (inject[2] is None and self.w.ncls > 1):
                raise ValueError("Engine.submit: inject = (dP, cellprob, logits) device tensors (logits may be None only without a class head)")
            if any(t is not None and (t.device != self.dev or not t.is_contiguous()) for t in inject):
                raise ValueError("Engine.submit: inject = (dP, cellprob, logits) contiguous tensors on the engine's device")
        sid = self._next
        self._next = (self._next + 1) % self.N_SLOTS
        sl = self.slots[sid]
        if sl.busy:
            raise RuntimeError("Engine: slot still holds an uncollected batch; call result() first")
        sl.n, sl.inject, sl.busy = n, inject, True
        cur = torch.cuda.current_stream(self.dev)
        self.s_net.wait_stream(cur)                     # inputs were produced on the caller's stream
        self.s_net.wait_event(sl.ev_post)               # this slot's previous batch left the head buffer
        lo_p, lo_g, hi_p, hi_g = self.pct
        sn, sp = self.s_net.cuda_stream, self.s_post.cuda_stream
        tm = self.stage_timing                          # bench.py's stage pass: event pairs around the stages the C-side profile does not see
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if tm is not None else None
        if ev:
            ev[0].record(self.s_net)
        check(self.L.cpx_normalize_stats_u8(ptr(tiles_u8), n, self.H, self.W, lo_p, lo_g, hi_p, hi_g,
                                            ptr(sl.stats), ptr(sl.hist), sn), "normalize_stats")
        check(self.L.cpx_make_patches(ptr(tiles_u8), ptr(sl.stats), n, C.byref(self.tiling), self.w.c.dtype,
                                      ptr(sl.patches), sn), "make_patches")
        if ev:
            ev[1].record(self.s_net)
        # the network always runs the full batch (rows of absent tiles hold old patches): kernel selection
        # and tile shapes depend on M, so a partial last batch would otherwise round differently --
        # this keeps a tile's outputs bitwise independent of batch composition, rank and world size
        check(self.L.cpx_net_forward(C.byref(self.w.c), ptr(sl.patches), self.nT * self.n_sub, ptr(sl.head),
                                     ptr(self.net_ws), self.net_ws_bytes, sn), "net_forward")
        if ev:
            ev[2].record(self.s_net)
        sl.ev_net.record(self.s_net)
        tiles_u8.record_stream(self.s_net)
        # post stream
        self.s_post.wait_event(sl.ev_net)
        if inject is not None:
            self.s_post.wait_stream(cur)
        ncls = self.w.ncls
        if ev:
            ev[3].record(self.s_post)
        check(self.L.cpx_blend_subtiles(ptr(sl.head), self.w.c.ld_head, ncls if ncls > 1 else 0, n,
                                        C.byref(self.tiling), ptr(self.taper), ptr(sl.dP),
                                        ptr(sl.cellprob), ptr(sl.logits), sp), "blend")
        if ev:
            ev[4].record(self.s_post)
            tm.append(ev)
        dP, cp, lg = (sl.dP, sl.cellprob, sl.logits) if inject is None else inject
        # one fused chain: ids, classes and (when asked for) the per-cell records leave in its last pass
        check(self.L.cpx_compute_masks_records(ptr(dP), ptr(cp), ptr(lg) if ncls > 1 else None, n, ncls,
                                               self.H, self.W, self.cp_thr, self.flow_thr, self.niter,
                                               self.min_size, self.max_frac, ptr(sl.masks), ptr(sl.class_masks),
                                               ptr(sl.nlabels), self.max_rec if records else 0,
                                               ptr(sl.records) if records else None, ptr(sl.rec_counts) if records else None,
                                               ptr(sl.pp_ws), sp), "compute_masks")
        sl.has_polygons = polygons is not None
        if polygons is not None:
            if not records:
                raise ValueError("Engine.submit: polygons need the per-cell records (records=True)")
            scale, origins = polygons
            if sl.cells is None:
                sl.cells = torch.empty(self.nT * self.max_rec * C.sizeof(_lib.CpxCell), dtype=torch.uint8, device=self.dev)
                sl.xy_pool = torch.empty((self.max_pts, 2), dtype=torch.float64, device=self.dev)
                # (torch.empty, not zeros: a fill kernel would run on the CALLER's stream, unordered against the post stream that writes
                # these buffers -- under GPU contention it landed after the H2D copy of the origins / after the vertex scan and
                # zeroed them: the first batch's cells at the slide origin, the 8-ranks-on-one-GPU test failing in 9 of 40 runs.
                # k_poly_scan writes n_pts_total, the copy below writes every row of origins)
                sl.n_pts_total = torch.empty(1, dtype=torch.int32, device=self.dev)
                sl.poly_ws = torch.empty(self.L.cpx_polygonize_workspace_bytes(self.nT, self.H, self.W, self.max_rec),
                                         dtype=torch.uint8, device=self.dev)
                sl.origins_host = torch.zeros((self.nT, 2), dtype=torch.float64).pin_memory()
                sl.origins = torch.empty((self.nT, 2), dtype=torch.float64, device=self.dev)
                self.s_post.wait_stream(cur)             # the allocator may have handed out blocks with work pending on the caller's stream
            sl.origins_host[:n] = torch.as_tensor(np.asarray(origins, dtype=np.float64).reshape(n, 2))
            with torch.cuda.stream(self.s_post):
                sl.origins.copy_(sl.origins_host, non_blocking=True)
            check(self.L.cpx_polygonize_device(ptr(sl.masks), ptr(sl.records), ptr(sl.rec_counts), n, self.H, self.W,
                                               self.max_rec, float(scale), ptr(sl.origins), ptr(sl.xy_pool),
                                               self.max_pts, ptr(sl.cells), ptr(sl.n_pts_total), ptr(sl.poly_ws), sp),
                  "polygonize_device")
        sl.ev_post.record(self.s_post)
        return sid

    def result(self, sid: int, wait: bool = True) -> "TileOutputs":
        """Outputs of a submitted batch; the caller's current stream is made to wait for them."""
        sl = self.slots[sid]
        if wait:
            torch.cuda.current_stream(self.dev).wait_event(sl.ev_post)
        sl.busy = False
        self._last = sl
        n = sl.n
        return TileOutputs(sl.masks[:n], sl.class_masks[:n], sl.nlabels[:n], sl.dP[:n], sl.cellprob[:n],
                           sl.logits[:n] if self.w.ncls > 1 else None, sl.records, sl.rec_counts,
                           *((sl.cells, sl.xy_pool, sl.n_pts_total) if sl.has_polygons else (None, None, None)))

    def run(self, tiles_u8: torch.Tensor, inject=None, records: bool = True, polygons=None) -> "TileOutputs":
        return self.result(self.submit(tiles_u8, inject, records, polygons))

    def fetch_polygons(self, n: int, out: "TileOutputs"):
        """Device polygons of a collected batch -> (cells CELL_DTYPE[m] with a ``tile`` column appended as a
        separate int array, xy float64 [total, 2]); None when the vertex pool overflowed (callers fall back to
        the host polygoniser for that batch)."""
        total = int(out.n_pts_total.item())
        if total > self.max_pts:
            return None
        counts = self._checked_counts(out.rec_counts, n)
        mx = int(counts.max()) if n else 0
        raw = out.cells.view(self.nT, self.max_rec, CELL_DTYPE.itemsize)[:n, :mx].cpu().numpy().view(CELL_DTYPE).reshape(n, mx)
        rows = [raw[t, :int(counts[t])] for t in range(n)]
        tile = np.concatenate([np.full(len(r), t, np.int32) for t, r in enumerate(rows)]) if n else np.zeros(0, np.int32)
        cells = np.concatenate(rows) if n else raw[:0, 0]
        return cells, tile, out.xy_pool[:total].cpu().numpy()

    # kept for the bench / CLI: records of the most recently collected batch
    @property
    def records(self):
        return self._last.records

    @property
    def rec_counts(self):
        return self._last.rec_counts

    def fetch_records(self, n: int, out: "TileOutputs | None" = None) -> np.ndarray:
        """Compact per-cell records of a collected batch as a structured numpy array (D2H)."""
        recs = self._last.records if out is None else out.records
        cnts = self._last.rec_counts if out is None else out.rec_counts
        counts = self._checked_counts(cnts, n)
        mx = int(counts.max()) if n else 0
        raw = recs.view(self.nT, self.max_rec, RECORD_DTYPE.itemsize)[:n, :mx].cpu().numpy().view(RECORD_DTYPE).reshape(n, mx)
        return np.concatenate([raw[t, :int(counts[t])] for t in range(n)]) if n else raw.reshape(-1)

    def _checked_counts(self, rec_counts: torch.Tensor, n: int) -> np.ndarray:
        counts = rec_counts[:n].cpu().numpy()
        if n and int(counts.max()) > self.max_rec:
            raise RuntimeError(f"a tile holds {int(counts.max())} instances but the record buffer has {self.max_rec} "
                               "slots per tile: cells would be dropped")
        return counts


CELL_DTYPE = np.dtype([("area", "<f8"), ("perimeter", "<f8"), ("cx", "<f8"), ("cy", "<f8"),
                       ("n_pts", "<i4"), ("offset", "<i4"), ("valid", "<i4"), ("cls", "<i4")])
assert CELL_DTYPE.itemsize == C.sizeof(_lib.CpxCell)
RECORD_DTYPE = np.dtype([("tile", "<i4"), ("label", "<i4"), ("cls", "<i4"), ("area", "<i4"),
                         ("y0", "<i4"), ("x0", "<i4"), ("y1", "<i4"), ("x1", "<i4"),
                         ("sum_y", "<i8"), ("sum_x", "<i8")])
assert RECORD_DTYPE.itemsize == C.sizeof(CpxRecord)
