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

from . import _lib, qc_arch as A
from ._lib import CpxQcOp, check, ptr

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
