"""GrandQC tissue / artefact detection on the MI355X engine (host side).

Mirrors ``detect_tissue_wsi`` (/root/reference/src/classpose/grandqc/wsi_tissue_detection.py:32-329),
``detect_artefacts_wsi`` (wsi_artefact_detection.py:56-348) and the helpers of
``wsi_qc_helpers.py``.  The networks run through ``cpx_qc_forward`` (float32, hand-written HIP,
``csrc/cpx_qc.hip``); this module flattens the smp/timm state dict into its operation list
(``QcNet``), cuts the thumbnail into 512-px patches exactly like the reference loops do, and does
the host-side raster post-processing (connected components, contours with holes, GeoJSON).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib, qc_arch as A
from .._lib import CpxQcOp, check, ptr

NONE = C.c_size_t(-1).value


def _pad16(c: int) -> int:
    return (c + 15) // 16 * 16


class _Tensor:
    """A float32 NHWC activation inside the workspace: channel slice [off, off + c) of rows of ld floats."""

    def __init__(self, base: int, ld: int, off: int, c: int, stride: int):
        self.base, self.ld, self.off, self.c, self.stride = base, ld, off, c, stride

    def byte_off(self) -> int:
        return (self.base + self.off) * 4


class QcNet:
    """``smp.UnetPlusPlus("timm-efficientnet-b0", classes=n)`` flattened for ``cpx_qc_forward``."""

    def __init__(self, sd: dict, device):
        self.n_classes = A.check_state_dict(sd)
        self.device = torch.device(device)
        self._keep: list[torch.Tensor] = []          # device weights (kept alive for the raw pointers)
        self._sd = {k: v.detach().to(torch.float64) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
        self._plans: dict = {}

    @classmethod
    def from_state_dict(cls, sd: dict, device) -> "QcNet":
        return cls(sd, device)

    # ---- weight preparation --------------------------------------------------------------
    def _dev(self, t: torch.Tensor) -> int:
        t = t.to(torch.float32).contiguous().to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def _bn_fold(self, bn: str | None, cout: int, conv_bias: torch.Tensor | None = None):
        if bn is None:
            scale = torch.ones(cout, dtype=torch.float64)
            shift = torch.zeros(cout, dtype=torch.float64) if conv_bias is None else conv_bias.clone()
            return scale, shift
        sd = self._sd
        scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + A.BN_EPS)
        shift = sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale
        if conv_bias is not None:
            shift = shift + conv_bias * scale
        return scale, shift

    def _dense_weights(self, wkey: str, bn: str | None, ca: int, cb: int, bias_key: str | None = None):
        """[Cout][Ca+Cb][k][k] -> device [CoutPad][k*k][pad16(Ca)+pad16(Cb)] (BN folded), bias [CoutPad]."""
        w = self._sd[wkey]
        cout, cin, k, _ = w.shape
        assert cin == ca + cb or (cin == 3 and ca == 4 and cb == 0), (wkey, cin, ca, cb)
        scale, shift = self._bn_fold(bn, cout, self._sd[bias_key] if bias_key else None)
        w = w * scale[:, None, None, None]
        tn = 32 if cout <= 32 else 64
        cpad = (cout + tn - 1) // tn * tn
        kc = _pad16(ca) + (_pad16(cb) if cb else 0)
        out = torch.zeros((cpad, k * k, kc), dtype=torch.float64)
        wt = w.permute(0, 2, 3, 1).reshape(cout, k * k, cin)
        na = min(ca, cin)
        out[:cout, :, :na] = wt[:, :, :na]
        if cb:
            out[:cout, :, _pad16(ca):_pad16(ca) + cb] = wt[:, :, ca:]
        b = torch.zeros(cpad, dtype=torch.float64)
        b[:cout] = shift
        return self._dev(out), self._dev(b)

    def _dw_weights(self, wkey: str, bn: str):
        w = self._sd[wkey]                                   # [C][1][k][k]
        c, _, k, _ = w.shape
        scale, shift = self._bn_fold(bn, c)
        wt = (w[:, 0] * scale[:, None, None]).permute(1, 2, 0).reshape(k * k, c)
        return self._dev(wt), self._dev(shift)

    # ---- plan: buffers + op list for (nB, H, W) --------------------------------------------
    def plan(self, nB: int, H: int, W: int):
        key = (nB, H, W)
        if key in self._plans:
            return self._plans[key]
        if H % 32 or W % 32:
            raise ValueError("GrandQC patches must be multiples of 32 px")
        cursor = [0]

        def alloc(stride: int, ld: int) -> int:
            base = cursor[0]
            cursor[0] += nB * (H // stride) * (W // stride) * ld
            cursor[0] = (cursor[0] + 63) // 64 * 64
            return base

        def alloc_flat(n: int) -> int:
            base = cursor[0]
            cursor[0] += (n + 63) // 64 * 64
            return base

        inp = _Tensor(alloc(1, 4), 4, 0, 4, 1)
        # UNet++ concat buffers per level (stride 2, 4, 8, 16): [x_1_l | ... | x_l_l | encoder feature]
        l1 = alloc(2, 128); l2 = alloc(4, 72); l3 = alloc(8, 80); l4 = alloc(16, 112)
        t = {
            "f4": _Tensor(l1, 128, 96, 32, 2), "x_1_3": _Tensor(l1, 128, 0, 32, 2),
            "x_2_3": _Tensor(l1, 128, 32, 32, 2), "x_3_3": _Tensor(l1, 128, 64, 32, 2),
            "f3": _Tensor(l2, 72, 48, 24, 4), "x_1_2": _Tensor(l2, 72, 0, 24, 4), "x_2_2": _Tensor(l2, 72, 24, 24, 4),
            "f2": _Tensor(l3, 80, 40, 40, 8), "x_1_1": _Tensor(l3, 80, 0, 40, 8),
            "f1": _Tensor(l4, 112, 0, 112, 16),
            "f0": _Tensor(alloc(32, 320), 320, 0, 320, 32),
            "x_0_0": _Tensor(alloc(16, 256), 256, 0, 256, 16), "x_0_1": _Tensor(alloc(8, 128), 128, 0, 128, 8),
            "x_0_2": _Tensor(alloc(4, 64), 64, 0, 64, 4), "x_0_3": _Tensor(alloc(2, 32), 32, 0, 32, 2),
            "x_0_4": _Tensor(alloc(1, 16), 16, 0, 16, 1),
        }
        blocks = A.encoder_blocks()
        n_exp = max(b["mid"] * (H // self._in_stride(b)) * (W // self._in_stride(b)) for b in blocks if b["kind"] == "ir")
        n_exp = max(n_exp, 16 * H * W)                        # also the decoder's conv1 temporary
        n_dw = max(b["mid"] * (H // self._out_stride(b)) * (W // self._out_stride(b)) for b in blocks)
        t_exp, t_dw = alloc_flat(nB * n_exp), alloc_flat(nB * n_dw)
        n_x = max(b["cout"] * (H // self._out_stride(b)) * (W // self._out_stride(b)) for b in blocks)
        xbuf = [alloc_flat(nB * n_x), alloc_flat(nB * n_x)]
        gate = alloc_flat(nB * 1152)
        pool = alloc_flat(16 * nB * 1152)
        ld_logits = (self.n_classes + 3) // 4 * 4
        logits = _Tensor(alloc(1, ld_logits), ld_logits, 0, self.n_classes, 1)

        ops: list[CpxQcOp] = []

        def conv(a: _Tensor, b: _Tensor | None, dst: _Tensor, wkey, bn, k, stride, act, up=0, gate_off=None,
                 res: _Tensor | None = None, bias_key=None):
            h_in = H // a.stride * (2 if up else 1)
            w_in = W // a.stride * (2 if up else 1)
            wp, bp = self._dense_weights(wkey, bn, a.c, b.c if b else 0, bias_key)
            o = CpxQcOp(kind=0, k=k, stride=stride, pad=k // 2, act=act, h_in=h_in, w_in=w_in,
                        h_out=h_in // stride, w_out=w_in // stride,
                        src_a=a.byte_off(), src_b=b.byte_off() if b else NONE,
                        gate=gate_off * 4 if gate_off is not None else NONE,
                        res=res.byte_off() if res else NONE, dst=dst.byte_off(),
                        c_a=a.c, ld_a=a.ld, up_a=up, c_b=b.c if b else 0, ld_b=b.ld if b else 0,
                        ld_res=res.ld if res else 0, c_out=dst.c, ld_dst=dst.ld, c_red=0,
                        w=wp, bias=bp, w2=None, bias2=None)
            ops.append(o)

        def dwconv(a: _Tensor, dst: _Tensor, wkey, bn, k, stride):
            wp, bp = self._dw_weights(wkey, bn)
            h_in, w_in = H // a.stride, W // a.stride
            ops.append(CpxQcOp(kind=1, k=k, stride=stride, pad=k // 2, act=2, h_in=h_in, w_in=w_in,
                               h_out=h_in // stride, w_out=w_in // stride, src_a=a.byte_off(), src_b=NONE, gate=NONE,
                               res=NONE, dst=dst.byte_off(), c_a=a.c, ld_a=a.ld, up_a=0, c_b=0, ld_b=0, ld_res=0,
                               c_out=a.c, ld_dst=a.c, c_red=0, w=wp, bias=bp, w2=None, bias2=None))

        def se(a: _Tensor, p: str, cr: int):
            sd = self._sd
            ops.append(CpxQcOp(kind=2, k=1, stride=1, pad=0, act=0, h_in=H // a.stride, w_in=W // a.stride,
                               h_out=1, w_out=1, src_a=a.byte_off(), src_b=NONE, gate=NONE, res=pool * 4,
                               dst=gate * 4, c_a=a.c, ld_a=a.ld, up_a=0, c_b=0, ld_b=0, ld_res=0, c_out=a.c,
                               ld_dst=a.c, c_red=cr,
                               w=self._dev(sd[p + "conv_reduce.weight"].reshape(cr, a.c)),
                               bias=self._dev(sd[p + "conv_reduce.bias"]),
                               w2=self._dev(sd[p + "conv_expand.weight"].reshape(a.c, cr)),
                               bias2=self._dev(sd[p + "conv_expand.bias"])))

        # ---- encoder
        conv(inp, None, t["f4"], "encoder.conv_stem.weight", "encoder.bn1", 3, 2, 2)
        x = t["f4"]
        feat_of_stage = {1: "f3", 2: "f2", 4: "f1", 6: "f0"}
        pp = 0
        for i, b in enumerate(blocks):
            p = b["prefix"]
            last = i + 1 == len(blocks) or blocks[i + 1]["stage"] != b["stage"]
            so = x.stride * b["stride"]
            if last and b["stage"] in feat_of_stage:
                dst = t[feat_of_stage[b["stage"]]]
            else:
                dst = _Tensor(xbuf[pp], b["cout"], 0, b["cout"], so)
                pp ^= 1
            if b["kind"] == "ds":
                d = _Tensor(t_dw, b["cin"], 0, b["cin"], so)
                dwconv(x, d, p + "conv_dw.weight", p + "bn1", b["k"], b["stride"])
                se(d, p + "se.", b["se"])
                conv(d, None, dst, p + "conv_pw.weight", p + "bn2", 1, 1, 0, gate_off=gate,
                     res=x if b["residual"] else None)
            else:
                e = _Tensor(t_exp, b["mid"], 0, b["mid"], x.stride)
                conv(x, None, e, p + "conv_pw.weight", p + "bn1", 1, 1, 2)
                d = _Tensor(t_dw, b["mid"], 0, b["mid"], so)
                dwconv(e, d, p + "conv_dw.weight", p + "bn2", b["k"], b["stride"])
                se(d, p + "se.", b["se"])
                conv(d, None, dst, p + "conv_pwl.weight", p + "bn3", 1, 1, 0, gate_off=gate,
                     res=x if b["residual"] else None)
            x = dst
        # ---- UNet++ decoder
        dec = A.decoder_blocks()
        for name, xsrc, skips in A.decoder_schedule():
            ci, cs, co = dec[name]
            a = t[xsrc]
            assert a.c == ci, (name, a.c, ci)
            dst = t[name]
            bsl = None
            if skips:
                first = t[skips[0]]
                bsl = _Tensor(first.base, first.ld, first.off, cs, first.stride)
                assert first.off + cs == first.ld and first.stride * 2 == a.stride, (name, skips)
            tmp = _Tensor(t_exp, co, 0, co, dst.stride)
            pfx = f"decoder.blocks.{name}."
            conv(a, bsl, tmp, pfx + "conv1.0.weight", pfx + "conv1.1", 3, 1, 1, up=1)
            conv(tmp, None, dst, pfx + "conv2.0.weight", pfx + "conv2.1", 3, 1, 1)
        conv(t["x_0_4"], None, logits, "segmentation_head.0.weight", None, 3, 1, 0,
             bias_key="segmentation_head.0.bias")
        arr = (CpxQcOp * len(ops))(*ops)
        plan = dict(ops=arr, n_ops=len(ops), ws_bytes=cursor[0] * 4, input_off=inp.byte_off(),
                    logits_off=logits.byte_off(), ld_logits=ld_logits,
                    ws=torch.empty(cursor[0] * 4, dtype=torch.uint8, device=self.device))
        self._plans = {key: plan}                            # one live plan (workspace) at a time
        return plan

    @staticmethod
    def _strides():
        s, out = 2, {}
        for b in A.encoder_blocks():
            out[(b["stage"], b["block"])] = (s, s * b["stride"])
            s *= b["stride"]
        return out

    def _in_stride(self, b):
        return self._strides()[(b["stage"], b["block"])][0]

    def _out_stride(self, b):
        return self._strides()[(b["stage"], b["block"])][1]

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, patches_u8: torch.Tensor, return_logits: bool = False):
        """patches_u8: (nB, H, W, 3) uint8 device tensor -> int8 class maps (nB, H, W) [, logits (nB,H,W,n)]."""
        assert patches_u8.dtype == torch.uint8 and patches_u8.dim() == 4 and patches_u8.shape[-1] == 3
        patches_u8 = patches_u8.contiguous()
        nB, H, W, _ = patches_u8.shape
        pl = self.plan(nB, H, W)
        cls = torch.empty((nB, H, W), dtype=torch.int8, device=self.device)
        logits = torch.empty((nB, H, W, self.n_classes), dtype=torch.float32, device=self.device) if return_logits else None
        check(_lib.lib().cpx_qc_forward(pl["ops"], pl["n_ops"], ptr(patches_u8), nB, H, W, pl["input_off"],
                                        pl["logits_off"], self.n_classes, pl["ld_logits"], ptr(cls),
                                        ptr(logits) if logits is not None else None, ptr(pl["ws"]), pl["ws_bytes"],
                                        torch.cuda.current_stream(self.device).cuda_stream), "qc_forward")
        return (cls, logits) if return_logits else cls


# =============================================================================================
# host pipeline
# =============================================================================================
import io
import os
import pickle
import uuid

from ..log import get_logger

grandqc_logger = get_logger("classpose.grandqc")

ARTIFACT_COLORS = [[0, 0, 0], [0, 0, 0], [255, 99, 71], [0, 255, 0], [255, 0, 0], [255, 0, 255], [75, 0, 130],
                   [255, 255, 255]]
ARTIFACT_CLASS_MAPPING = {0: "Unused", 1: "Normal Tissue", 2: "Fold", 3: "Darkspot & Foreign Object",
                          4: "PenMarking", 5: "Edge & Air Bubble", 6: "OOF", 7: "Background"}
QC_BATCH = 8


# ---- checkpoints ------------------------------------------------------------------------------
class _Stub:
    """Stand-in for smp / timm classes of a pickled module: keeps the instance state only."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


class _StateOnlyUnpickler(pickle.Unpickler):
    """Reads a pickled ``nn.Module`` tree (``torch.save(model)``, wsi_artefact_detection.py:124) for its tensors
    only.  ``find_class`` resolves an EXPLICIT allow-list of (module, name) pairs -- the tensor / storage
    rebuild helpers, dtypes, ``OrderedDict``, the numpy array reconstructors, ``copyreg``'s object
    constructors and a few value builtins; every other global (smp / timm / torch.nn classes,
    ``functools.partial``, anything else) becomes an inert ``_Stub`` subclass that only keeps instance state.
    No attribute lookup (``getattr``), no callable of the defining packages and no whole-module prefix is
    reachable from the pickle stream."""
    ALLOWED = {
        ("collections", "OrderedDict"), ("collections", "defaultdict"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
        ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
        ("torch._utils", "_rebuild_qtensor"), ("torch._tensor", "_rebuild_from_type_v2"),
        ("torch.nn.parameter", "Parameter"), ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
        ("torch.serialization", "_get_layout"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"),
        ("copyreg", "_reconstructor"), ("copyreg", "__newobj__"),
    }
    SAFE_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes",
                     "bytearray", "complex", "slice", "range", "object"}

    def find_class(self, module, name):
        if (module, name) in self.ALLOWED:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)):
            return super().find_class(module, name)              # torch.FloatStorage ..., torch.float32 ...
        if module == "builtins":
            if name in self.SAFE_BUILTINS:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"refusing builtins.{name} in a model checkpoint")
        if module.split(".")[0] in ("os", "posix", "nt", "subprocess", "sys", "importlib", "runpy", "shutil", "socket"):
            raise pickle.UnpicklingError(f"refusing {module}.{name} in a model checkpoint")
        return type(name, (_Stub,), {"__module__": module})


class _StatePickle:
    __name__ = "classpose_amd_state_pickle"
    Unpickler = _StateOnlyUnpickler

    @staticmethod
    def load(f, **kw):
        return _StateOnlyUnpickler(f, **kw).load()


def _module_tree_to_state_dict(obj, prefix: str = "", out: dict | None = None) -> dict:
    out = {} if out is None else out
    d = getattr(obj, "__dict__", {})
    for name, p in (d.get("_parameters") or {}).items():
        if p is not None:
            out[prefix + name] = p.detach()
    for name, b in (d.get("_buffers") or {}).items():
        if b is not None:
            out[prefix + name] = b.detach()
    for name, m in (d.get("_modules") or {}).items():
        if m is not None:
            _module_tree_to_state_dict(m, prefix + name + ".", out)
    return out


def load_qc_state_dict(path: str, n_classes: int, seed: int) -> dict:
    """Tissue model: ``torch.save(state_dict)``; artefact model: a pickled smp module.  Both come
    back as a flat state dict with smp's key names.  Without the file (no network here) seeded
    synthetic weights are used when CLASSPOSE_SYNTHETIC_WEIGHTS=1, else FileNotFoundError."""
    if not os.path.exists(path):
        if os.getenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "0") == "1":
            from .. import synth
            grandqc_logger.warning(f"No weights at {path}: using seeded synthetic GrandQC weights")
            return synth.make_grandqc_state_dict(n_classes, seed)
        raise FileNotFoundError(f"{path} not found (downloads are not available: fetch the GrandQC checkpoint "
                                "manually, see the reference's MODEL_URL_PATH)")
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        grandqc_logger.warning(f"{path} is not a plain state dict ({type(e).__name__}): reading it as a pickled module "
                               "through the state-only unpickler (allow-listed globals, no package code runs)")
        obj = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_StatePickle)
    if isinstance(obj, dict):
        sd = obj.get("state_dict", obj)
        return {k: v for k, v in sd.items() if torch.is_tensor(v)}
    return _module_tree_to_state_dict(obj)


# ---- helpers (wsi_qc_helpers.py) ----------------------------------------------------------------
def simulate_jpeg_compression(image: np.ndarray) -> np.ndarray:
    """JPEG quality-80 round trip (wsi_qc_helpers.py:7-23).  The reference hands the RGB array to
    ``cv2.imencode``, which reads it as BGR: the codec sees red and blue swapped.  PIL (libjpeg,
    4:2:0, same quality scaling) stands in for OpenCV's codec, with the same channel swap."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(image[..., ::-1])).save(buf, format="JPEG", quality=80)
    buf.seek(0)
    return np.ascontiguousarray(np.asarray(Image.open(buf).convert("RGB"))[..., ::-1])


def extract_slide_info(slide, mpp_model: float):
    from ..wsi import get_slide_resolution
    w_l0, h_l0 = slide.level_dimensions[0]
    mpp = get_slide_resolution(slide)[0]
    reduction_factor = mpp_model / mpp
    return w_l0, h_l0, mpp, (int(w_l0 // reduction_factor), int(h_l0 // reduction_factor))


def _thumbnail_rgb(slide, dims) -> np.ndarray:
    img = slide.get_thumbnail(dims)
    if not isinstance(img, np.ndarray):
        img = np.asarray(img.convert("RGB"))
    return np.ascontiguousarray(img[..., :3])


def make_class_map(mask: np.ndarray, class_colors: list[list[int]]) -> np.ndarray:
    """wsi_qc_helpers.make_class_map: class index -> RGB"""
    lut = np.zeros((256, 3), np.uint8)
    lut[: len(class_colors)] = np.asarray(class_colors, dtype=np.uint8)
    return lut[np.asarray(mask).astype(np.uint8)]


def draw_contour_outlines(shape, contours, thickness: int = 10) -> np.ndarray:
    """``cv2.drawContours(img, [cnt], 0, 255, thickness=10)`` for the diagnostic ``filled_class_map``
    image (wsi_tissue_detection.py:239): closed polylines of the given thickness, drawn with PIL
    (OpenCV's exact line rasteriser is not restated; nothing downstream reads this image)."""
    from PIL import Image, ImageDraw
    img = Image.new("L", (shape[1], shape[0]), 0)
    d = ImageDraw.Draw(img)
    for c in contours:
        pts = [tuple(map(float, p)) for p in np.asarray(c)]
        if len(pts) >= 2:
            d.line(pts + [pts[0]], fill=255, width=thickness, joint="curve")
    return np.asarray(img)


def resize_nearest(mask: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(mask, (width, height), interpolation=cv2.INTER_NEAREST): src = min(floor(dst * scale), n-1)"""
    sh, sw = mask.shape[:2]
    ys = np.minimum(np.floor(np.arange(height) * (1.0 / (height / sh))).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(width) * (1.0 / (width / sw))).astype(np.int64), sw - 1)
    return mask[ys][:, xs]


def find_contours_ccomp(mask: np.ndarray):
    """cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_SIMPLE) -> (list of (n, 2) int32 arrays, parent index
    array); host C++ (``cpx_find_contours_ccomp_host``)."""
    m = np.ascontiguousarray(mask != 0, dtype=np.uint8)
    H, W = m.shape
    max_pts, max_c = max(4096, 2 * int(m.sum()) + 16), max(1024, int(m.sum()) // 2 + 16)
    while True:
        xy = np.empty((max_pts, 2), np.int32)
        offs, npts, par = (np.empty(max_c, np.int32) for _ in range(3))
        n = _lib.lib().cpx_find_contours_ccomp_host(m.ctypes.data, H, W, xy.ctypes.data, max_pts, offs.ctypes.data,
                                                    npts.ctypes.data, par.ctypes.data, max_c)
        if n == -12:                                        # CPX_ENOMEM: grow and retry
            max_pts, max_c = max_pts * 2, max_c * 2
            continue
        if n < 0:
            raise _lib.CpxError(f"cpx_find_contours_ccomp_host failed ({n})")
        return [xy[offs[i]: offs[i] + npts[i]].copy() for i in range(n)], par[:n].copy()


def contour_area(cnt: np.ndarray) -> float:
    """cv2.contourArea: |shoelace| / 2"""
    x, y = cnt[:, 0].astype(np.float64), cnt[:, 1].astype(np.float64)
    return abs(float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))) / 2.0


def _patch_specs(width: int, height: int, p: int):
    """the reference's (he_n + 1) x (wi_n + 1) loop: full grid + re-anchored last row / column
    (wsi_tissue_detection.py:133-151).  Yields (h, w, crop box)."""
    wi_n, he_n = width // p, height // p
    for h in range(he_n + 1):
        for w in range(wi_n + 1):
            x0 = w * p if w != wi_n else width - p
            y0 = h * p if h != he_n else height - p
            yield h, w, (x0, y0, x0 + p, y0 + p)


def _crop_pil_like(image: np.ndarray, box) -> np.ndarray:
    """PIL Image.crop: regions outside the image are black"""
    x0, y0, x1, y1 = box
    H, W = image.shape[:2]
    out = np.zeros((y1 - y0, x1 - x0, 3), np.uint8)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0: sy1 - y0, sx0 - x0: sx1 - x0] = image[sy0:sy1, sx0:sx1]
    return out


def _run_patches(net: QcNet, patches: list[np.ndarray]) -> list[np.ndarray]:
    out = []
    for s in range(0, len(patches), QC_BATCH):
        chunk = np.stack(patches[s: s + QC_BATCH])
        cls = net.forward(torch.from_numpy(chunk).to(net.device))
        out.extend(cls.cpu().numpy())
    return out


def _as_net(model, n_classes: int, device, seed: int) -> QcNet:
    if isinstance(model, QcNet):
        return model
    sd = model if isinstance(model, dict) else load_qc_state_dict(str(model), n_classes, seed)
    return QcNet.from_state_dict(sd, device)


# ---- detect_tissue_wsi ------------------------------------------------------------------------
def tissue_class_map(image: np.ndarray, net: QcNet, p_s: int = 512) -> np.ndarray:
    """patch loop + assembly of detect_tissue_wsi (:131-196): int8 class map of the thumbnail"""
    height, width = image.shape[:2]
    wi_n, he_n = width // p_s, height // p_s
    over_w, over_h = width - wi_n * p_s, height - he_n * p_s
    specs = list(_patch_specs(width, height, p_s))
    # a re-anchored edge patch only contributes its last `overhang` columns / rows: skip it when that is nothing
    used = [(h, w, box) for h, w, box in specs if not ((w == wi_n and over_w == 0) or (h == he_n and over_h == 0))]
    masks = _run_patches(net, [_crop_pil_like(image, box) for _, _, box in used])
    out = np.zeros((height, width), np.int8)
    for (h, w, box), m in zip(used, masks):
        ys = slice(p_s - over_h, p_s) if h == he_n else slice(0, p_s)
        xs = slice(p_s - over_w, p_s) if w == wi_n else slice(0, p_s)
        y0 = height - over_h if h == he_n else h * p_s
        x0 = width - over_w if w == wi_n else w * p_s
        sub = m[ys, xs]
        out[y0: y0 + sub.shape[0], x0: x0 + sub.shape[1]] = sub
    return out


def tissue_contours(class_map: np.ndarray, mpp_model_td: int, min_area: int, scaling):
    """connected components of class 0 (tissue) with the real-area filter, then RETR_CCOMP contours
    with holes attached to their parents (:198-250).  Returns (filtered_mask, output_cnts)."""
    from scipy import ndimage
    fg = (1 - class_map.astype(np.uint8)).astype(np.uint8)      # cv2.connectedComponents(1 - map): tissue = class 0
    cc, n_c = ndimage.label(fg != 0, structure=np.ones((3, 3), int))
    filtered = np.zeros_like(fg, dtype=np.uint8)
    if n_c:
        areas = np.bincount(cc.ravel(), minlength=n_c + 1) * (mpp_model_td ** 2)
        keep = areas >= min_area
        keep[0] = False
        filtered[keep[cc]] = 1
    cnts, parent = find_contours_ccomp(filtered)
    output_cnts: dict = {}
    scaling = np.asarray(scaling, dtype=np.float64)
    for i, cnt in enumerate(cnts):
        if cnt.shape[0] < 4:
            grandqc_logger.warning(f"Invalid polygon detected: fewer than 4 points detected ({cnt.shape})")
            continue
        if parent[i] == -1:
            c = cnt * scaling
            output_cnts[i] = {"contour": np.concatenate([c, c[0:1]], 0), "holes": []}
    for i in np.nonzero(parent != -1)[0]:
        if int(parent[i]) in output_cnts:                       # the reference raises KeyError here when the parent was < 4 points
            output_cnts[int(parent[i])]["holes"].append(cnts[i] * scaling)
    return filtered, output_cnts


def _cnts_to_geojson(output_cnts: dict, name: str, color: list[int]) -> dict:
    feats = []
    for cnt in output_cnts.values():
        coords = cnt["contour"].tolist()
        coords.append(coords[0])
        rings = []
        for hole in cnt["holes"]:
            hc = hole.tolist()
            if len(hc) < 4:
                continue
            if hc[0] != hc[-1]:
                hc.append(hc[0])
            rings.append(hc)
        feats.append({"type": "Feature", "id": str(uuid.uuid4()),
                      "geometry": {"type": "Polygon", "coordinates": [coords, *rings]},
                      "properties": {"objectType": "annotation", "isLocked": False,
                                     "classification": {"name": name, "color": color}}})
    return {"type": "FeatureCollection", "features": feats}


def _shift_outputs(output_cnts: dict, geojson: dict, bx: float, by: float):
    off = np.array([bx, by])
    for cnt in output_cnts.values():
        cnt["contour"] = cnt["contour"] - off
        cnt["holes"] = [h - off for h in cnt["holes"]]
    for f in geojson["features"]:
        f["geometry"]["coordinates"] = [[[pt[0] - bx, pt[1] - by] for pt in ring]
                                        for ring in f["geometry"]["coordinates"]]


def detect_tissue_wsi(slide, model_td_path="./models/tissue_detection/Tissue_Detection_MPP10.pth",
                      mpp_model_td: int = 10, m_p_s_model_td: int = 512, device="cuda:0", min_area: int = 0,
                      apply_bounds_offset: bool = False, class_map_override=None):
    """Same return tuple as the reference: (image, filtered_mask, filled_class_map, output_cnts,
    geojson, mpp_model_td).  ``filled_class_map`` is the contour-outline rendering (thickness 10) of
    ``draw_contour_outlines``.  ``class_map_override(image) -> int8 map`` replaces the network's
    decision in flow-injection style tests (the network still runs)."""
    net = _as_net(model_td_path, 2, device, seed=101)
    bx = float(slide.properties.get("openslide.bounds-x", 0.0))
    by = float(slide.properties.get("openslide.bounds-y", 0.0))
    w_l0, h_l0, mpp, dims = extract_slide_info(slide, mpp_model_td)
    grandqc_logger.info(f"Extracting thumbnail with size {dims}")
    image = simulate_jpeg_compression(_thumbnail_rgb(slide, dims))
    height, width = image.shape[:2]
    class_map = tissue_class_map(image, net, m_p_s_model_td)
    if class_map_override is not None:
        class_map = np.asarray(class_map_override(image), dtype=np.int8)
    filtered, output_cnts = tissue_contours(class_map, mpp_model_td, min_area, (w_l0 / width, h_l0 / height))
    filled = draw_contour_outlines(filtered.shape, [c["contour"] / np.array([w_l0 / width, h_l0 / height])
                                                    for c in output_cnts.values()], 10)
    if not output_cnts:
        grandqc_logger.warning("No tissue contours detected in slide.")
        return image, filtered, filled, {}, {"type": "FeatureCollection", "features": []}, mpp_model_td
    geojson = _cnts_to_geojson(output_cnts, "tissue", [0, 0, 0])
    if apply_bounds_offset and (bx != 0 or by != 0):
        _shift_outputs(output_cnts, geojson, bx, by)
    return image, filtered, filled, output_cnts, geojson, mpp_model_td


# ---- detect_artefacts_wsi ---------------------------------------------------------------------
def artefact_class_map(image: np.ndarray, tissue_mask_art: np.ndarray, net: QcNet, p_s: int = 512,
                       class_map_override=None) -> np.ndarray:
    """tissue-gated patch loop + padding of detect_artefacts_wsi (:175-229): class 7 = background"""
    height, width = image.shape[:2]
    n_w, n_h = width // p_s, height // p_s
    out = np.full((height, width), 7, dtype=np.int64)
    todo = []
    for h in range(n_h):
        for w in range(n_w):
            td = tissue_mask_art[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s]
            if np.count_nonzero(td == 1) > 50:
                todo.append((h, w))
    masks = _run_patches(net, [image[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s] for h, w in todo])
    if class_map_override is not None:
        full = np.asarray(class_map_override(image))
        masks = [full[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s] for h, w in todo]
    for (h, w), m in zip(todo, masks):
        td = tissue_mask_art[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s]
        out[h * p_s:(h + 1) * p_s, w * p_s:(w + 1) * p_s] = np.where(td == 1, m, 7)
    return out


def create_geojson_feature(contour_points: np.ndarray, scaling_factors, classification_name: str,
                           classification_color: list[int]):
    """wsi_qc_helpers.create_geojson_feature (:49-88): contour x scale -> closed ring -> QuPath annotation feature;
    None for fewer than 4 points."""
    scaled = np.asarray(contour_points) * np.asarray(scaling_factors)
    if len(scaled) < 4:
        return None
    pts = scaled.tolist()
    if pts[0] != pts[-1]:
        pts.append(pts[0])
    return {"type": "Feature", "id": str(uuid.uuid4()),
            "geometry": {"type": "Polygon", "coordinates": [pts]},
            "properties": {"objectType": "annotation", "isLocked": False,
                           "classification": {"name": classification_name, "color": classification_color}}}


def artefact_contours(artefact_mask: np.ndarray, scaling):
    """per-class RETR_CCOMP contours (:252-325): GeoJSON features for classes 1-6, filter polygons
    (``artefact_cnts``) for classes 2-6 with area > 10 px"""
    scaling = np.asarray(scaling, dtype=np.float64)
    geojson = {"type": "FeatureCollection", "features": []}
    artefact_cnts: dict = {}
    n_small = 0
    for cv in range(1, 7):
        cnts, parent = find_contours_ccomp((artefact_mask == cv).astype(np.uint8))
        if not cnts:
            continue
        for i, cnt in enumerate(cnts):
            if cnt.shape[0] < 4:
                continue
            if cv >= 2 and contour_area(cnt) <= 10:
                n_small += 1
                continue
            geojson["features"].append(create_geojson_feature(cnt, scaling, ARTIFACT_CLASS_MAPPING.get(cv, "Unknown"),
                                                              ARTIFACT_COLORS[cv]))
            if 2 <= cv <= 6 and parent[i] == -1:
                c = cnt * scaling
                artefact_cnts[f"{cv}_{i}"] = {"contour": np.concatenate([c, c[0:1]], 0), "holes": []}
        if 2 <= cv <= 6:
            for i in np.nonzero(parent != -1)[0]:
                key = f"{cv}_{int(parent[i])}"
                if key in artefact_cnts:
                    artefact_cnts[key]["holes"].append(cnts[i] * scaling)
    grandqc_logger.info(f"Filtered {n_small} small artifacts (<= 10 pixels)")
    return artefact_cnts, geojson


def detect_artefacts_wsi(slide, model_art_path="./models/artefact_detection/GrandQC_MPP1.pth",
                         mpp_model_art: float = 1.0, m_p_s_model_art: int = 512, device="cuda:0",
                         model_td_path="./models/tissue_detection/Tissue_Detection_MPP10.pth", mpp_model_td: int = 10,
                         m_p_s_model_td: int = 512, min_area: int = 0, apply_bounds_offset: bool = False,
                         tissue_override=None, artefact_override=None):
    """(artefact_mask, artefact_map, artefact_cnts, geojson) like the reference; ``artefact_map`` is
    the colour rendering resized to 50 px per patch with PIL LANCZOS (:232-239)."""
    grandqc_logger.info("Performing tissue detection...")
    _, tissue_mask, _, _, _, _ = detect_tissue_wsi(slide, model_td_path, mpp_model_td, m_p_s_model_td, device,
                                                   min_area, False, class_map_override=tissue_override)
    net = _as_net(model_art_path, 8, device, seed=202)
    bx = float(slide.properties.get("openslide.bounds-x", 0.0))
    by = float(slide.properties.get("openslide.bounds-y", 0.0))
    w_l0, h_l0, mpp, dims = extract_slide_info(slide, mpp_model_art)
    grandqc_logger.info(f"Extracting thumbnail with size {dims} for artifact detection")
    image = simulate_jpeg_compression(_thumbnail_rgb(slide, dims))
    height, width = image.shape[:2]
    tissue_mask_art = resize_nearest(tissue_mask, width, height)
    artefact_mask = artefact_class_map(image, tissue_mask_art, net, m_p_s_model_art, artefact_override)
    artefact_cnts, geojson = artefact_contours(artefact_mask, (w_l0 / width, h_l0 / height))
    if apply_bounds_offset and (bx != 0 or by != 0):
        _shift_outputs(artefact_cnts, geojson, bx, by)
    from PIL import Image
    amap = Image.fromarray(make_class_map(artefact_mask, ARTIFACT_COLORS)).resize(
        (max(1, int(width * 50 / m_p_s_model_art)), max(1, int(height * 50 / m_p_s_model_art))), Image.Resampling.LANCZOS)
    return artefact_mask, np.array(amap), artefact_cnts, geojson
