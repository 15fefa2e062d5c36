"""Thin torch-tensor wrappers over the C ABI, one per reference call it replaces.

Names follow the reference / cellpose functions (``follow_flows``,
``get_masks``, ``remove_bad_flow_masks``, ``fill_holes_and_remove_small_masks``,
``compute_class_masks``, ``remove_border_instances``, ``normalize_img`` ...) so
the parity tests read like the reference's own tests.  All tensors live on the
GPU; every function raises if the HIP library is missing (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import CpxTiling, check, ptr
from .engine import make_tiling, percentile_params, taper_1d

_ws_cache: dict = {}


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _pp_ws(nT: int, H: int, W: int, dev) -> torch.Tensor:
    key = ("pp", nT, H, W, str(dev))
    if key not in _ws_cache:
        n = _lib.lib().cpx_postproc_workspace_bytes(nT, H, W)
        _ws_cache[key] = torch.empty(n, dtype=torch.uint8, device=dev)
    return _ws_cache[key]


def _batched(x: torch.Tensor, nd: int):
    """Accept the un-batched reference shape; return (batched view, was_batched)."""
    if x.dim() == nd:
        return x.unsqueeze(0), False
    return x, True


# ---- a6 -------------------------------------------------------------------
def normalize_stats(tiles_u8: torch.Tensor) -> torch.Tensor:
    tiles_u8, _ = _batched(tiles_u8, 3)
    nT, H, W, _c = tiles_u8.shape
    dev = tiles_u8.device
    stats = torch.empty((nT, 3, 4), dtype=torch.float32, device=dev)
    hist = torch.empty(nT * 768, dtype=torch.int32, device=dev)
    lo, hi = percentile_params(H * W, 1), percentile_params(H * W, 99)
    check(_lib.lib().cpx_normalize_stats_u8(ptr(tiles_u8), nT, H, W, lo[0], lo[1], hi[0], hi[1],
                                            ptr(stats), ptr(hist), _stream(dev)), "normalize_stats")
    return stats


def normalize_img(tiles_u8: torch.Tensor) -> torch.Tensor:
    """cellpose transforms.normalize_img on uint8 (n,H,W,3) tiles -> float32 (n,H,W,3)."""
    t, was = _batched(tiles_u8.contiguous(), 3)
    stats = normalize_stats(t)
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    check(_lib.lib().cpx_normalize_apply_u8(ptr(t), ptr(stats), t.shape[0], t.shape[1], t.shape[2],
                                            ptr(out), _stream(t.device)), "normalize_apply")
    return out if was else out[0]


# ---- a3 -------------------------------------------------------------------
def resized_shape(h: int, w: int, resize_factor: float) -> tuple[int, int]:
    """Output (h, w) of resize_tile_to_target_mpp (predict_wsi.py:117-118)."""
    return max(1, int(round(h * resize_factor))), max(1, int(round(w * resize_factor)))


def resize_tile_to_target_mpp(tiles_u8: torch.Tensor, resize_factor: float,
                              out: torch.Tensor | None = None) -> torch.Tensor:
    """predict_wsi.resize_tile_to_target_mpp on uint8 (n,h,w,3) / (h,w,3) device tiles:
    cv2.resize(..., INTER_LINEAR) to round(h*f) x round(w*f); factor 1.0 returns the input."""
    if resize_factor == 1.0:
        return tiles_u8
    t, was = _batched(tiles_u8.contiguous(), 3)
    dh, dw = resized_shape(t.shape[1], t.shape[2], resize_factor)
    if out is None:
        out = torch.empty((t.shape[0], dh, dw, 3), dtype=torch.uint8, device=t.device)
    check(_lib.lib().cpx_resize_linear_u8(ptr(t), t.shape[0], t.shape[1], t.shape[2], ptr(out), dh, dw,
                                          _stream(t.device)), "resize_linear_u8")
    return out if was else out[0]


# ---- a7 -------------------------------------------------------------------
def make_subtiles(tiles_u8: torch.Tensor, bsize: int = 256, augment: bool = False,
                  tile_overlap: float = 0.1):
    """pad + make_tiles of core.run_net on normalised pixels -> float32 (n*ny*nx, 3, b, b)."""
    t, _ = _batched(tiles_u8.contiguous(), 3)
    nT, H, W, _c = t.shape
    til = make_tiling(H, W, bsize, augment, tile_overlap)
    stats = normalize_stats(t)
    out = torch.empty((nT * til.ny * til.nx, 3, bsize, bsize), dtype=torch.float32, device=t.device)
    check(_lib.lib().cpx_make_subtiles_f32(ptr(t), ptr(stats), nT, C.byref(til), ptr(out),
                                           _stream(t.device)), "make_subtiles_f32")
    return out, til


def make_patches(tiles_u8: torch.Tensor, bsize: int = 256, augment: bool = False,
                 tile_overlap: float = 0.1, dtype: torch.dtype = torch.bfloat16):
    t, _ = _batched(tiles_u8.contiguous(), 3)
    nT, H, W, _c = t.shape
    til = make_tiling(H, W, bsize, augment, tile_overlap)
    stats = normalize_stats(t)
    nS = nT * til.ny * til.nx
    out = torch.empty((nS * (bsize // 8) ** 2, 192), dtype=dtype, device=t.device)
    check(_lib.lib().cpx_make_patches(ptr(t), ptr(stats), nT, C.byref(til), _DT[dtype], ptr(out),
                                      _stream(t.device)), "make_patches")
    return out, til


def blend_subtiles(y: torch.Tensor, y_class: torch.Tensor, til: CpxTiling, nT: int):
    """unaugment + average_tiles + crop of core.run_net: y (nS,3,b,b), y_class (nS,ncls,b,b)."""
    dev = y.device
    ncls = y_class.shape[1]
    H, W = til.H, til.W
    dP = torch.empty((nT, 2, H, W), dtype=torch.float32, device=dev)
    cp = torch.empty((nT, H, W), dtype=torch.float32, device=dev)
    lg = torch.empty((nT, ncls, H, W), dtype=torch.float32, device=dev)
    taper = torch.from_numpy(taper_1d(til.bsize)).to(dev)
    check(_lib.lib().cpx_blend_subtiles_nchw(ptr(y.contiguous()), ptr(y_class.contiguous()), ncls, nT,
                                             C.byref(til), ptr(taper), ptr(dP), ptr(cp), ptr(lg),
                                             _stream(dev)), "blend_nchw")
    return dP, cp, lg


def blend_head(head: torch.Tensor, ld_head: int, ncls: int, til: CpxTiling, nT: int):
    dev = head.device
    H, W = til.H, til.W
    dP = torch.empty((nT, 2, H, W), dtype=torch.float32, device=dev)
    cp = torch.empty((nT, H, W), dtype=torch.float32, device=dev)
    lg = torch.empty((nT, max(ncls, 1), H, W), dtype=torch.float32, device=dev)
    taper = torch.from_numpy(taper_1d(til.bsize)).to(dev)
    check(_lib.lib().cpx_blend_subtiles(ptr(head), ld_head, ncls, nT, C.byref(til), ptr(taper),
                                        ptr(dP), ptr(cp), ptr(lg), _stream(dev)), "blend")
    return dP, cp, lg


# ---- a11-a16 ----------------------------------------------------------------
def follow_flows(dP: torch.Tensor, cellprob: torch.Tensor, niter: int = 200,
                 cellprob_threshold: float = 0.0, return_float: bool = False):
    """dP (n,2,H,W) RAW network flows, cellprob (n,H,W).  Returns packed int32 end
    points (n,H*W) [(y<<16)|x, -1 = inactive] and optionally float (n,2,H*W)."""
    dP, _ = _batched(dP.contiguous(), 3)
    cellprob, _ = _batched(cellprob.contiguous(), 2)
    nT, _two, H, W = dP.shape
    dev = dP.device
    pf = torch.empty((nT, H * W), dtype=torch.int32, device=dev)
    fl = torch.empty((nT, 2, H * W), dtype=torch.float32, device=dev) if return_float else None
    check(_lib.lib().cpx_follow_flows(ptr(dP), ptr(cellprob), nT, H, W, cellprob_threshold, niter,
                                      ptr(pf), ptr(fl), ptr(_pp_ws(nT, H, W, dev)), _stream(dev)),
          "follow_flows")
    return (pf, fl) if return_float else pf


def get_masks(p_final: torch.Tensor, H: int, W: int, max_size_fraction: float = 0.4):
    nT = p_final.shape[0]
    dev = p_final.device
    masks = torch.empty((nT, H, W), dtype=torch.int32, device=dev)
    nlab = torch.empty(nT, dtype=torch.int32, device=dev)
    check(_lib.lib().cpx_get_masks(ptr(p_final.contiguous()), nT, H, W, max_size_fraction, ptr(masks),
                                   ptr(nlab), ptr(_pp_ws(nT, H, W, dev)), _stream(dev)), "get_masks")
    return masks, nlab


def remove_bad_flow_masks(masks: torch.Tensor, dP: torch.Tensor, threshold: float = 0.4,
                          return_errors: bool = False):
    """In place on int32 masks (n,H,W); dP (n,2,H,W) raw network flows."""
    nT, H, W = masks.shape
    dev = masks.device
    L = _lib.lib().cpx_postproc_max_labels(H, W)
    errs = torch.zeros((nT, L), dtype=torch.float64, device=dev) if return_errors else None
    check(_lib.lib().cpx_remove_bad_flow_masks(ptr(masks), ptr(dP.contiguous()), nT, H, W, threshold,
                                               ptr(errs), ptr(_pp_ws(nT, H, W, dev)), _stream(dev)),
          "remove_bad_flow_masks")
    return (masks, errs) if return_errors else masks


def fill_holes_and_remove_small_masks(masks: torch.Tensor, min_size: int = 15):
    nT, H, W = masks.shape
    dev = masks.device
    nlab = torch.empty(nT, dtype=torch.int32, device=dev)
    check(_lib.lib().cpx_fill_holes_and_remove_small_masks(ptr(masks), nT, H, W, min_size, ptr(nlab),
                                                           ptr(_pp_ws(nT, H, W, dev)), _stream(dev)),
          "fill_holes_and_remove_small_masks")
    return masks, nlab


def compute_class_masks(masks: torch.Tensor, y_class: torch.Tensor) -> torch.Tensor:
    """masks int32 (n,H,W), y_class float32 (n,ncls,H,W) -> uint8 (n,H,W)."""
    nT, H, W = masks.shape
    dev = masks.device
    cm = torch.empty((nT, H, W), dtype=torch.uint8, device=dev)
    check(_lib.lib().cpx_compute_class_masks(ptr(masks), ptr(y_class.contiguous()), nT, y_class.shape[1],
                                             H, W, ptr(cm), ptr(_pp_ws(nT, H, W, dev)), _stream(dev)),
          "compute_class_masks")
    return cm


def remove_border_instances(masks: torch.Tensor, class_masks: torch.Tensor | None = None):
    nT, H, W = masks.shape
    dev = masks.device
    check(_lib.lib().cpx_remove_border_instances(ptr(masks), ptr(class_masks), nT, H, W,
                                                 ptr(_pp_ws(nT, H, W, dev)), _stream(dev)),
          "remove_border_instances")
    return masks if class_masks is None else (masks, class_masks)


def compute_masks(dP: torch.Tensor, cellprob: torch.Tensor, logits: torch.Tensor | None = None,
                  niter: int = 200, cellprob_threshold: float = 0.0, flow_threshold: float = 0.4,
                  min_size: int = 15, max_size_fraction: float = 0.4):
    """dynamics.resize_and_compute_masks (+ compute_class_masks) on a batch of tiles."""
    nT, _two, H, W = dP.shape
    dev = dP.device
    ncls = 0 if logits is None else logits.shape[1]
    masks = torch.empty((nT, H, W), dtype=torch.int16, device=dev)
    cm = torch.empty((nT, H, W), dtype=torch.uint8, device=dev)
    nlab = torch.empty(nT, dtype=torch.int32, device=dev)
    check(_lib.lib().cpx_compute_masks(ptr(dP.contiguous()), ptr(cellprob.contiguous()),
                                       ptr(logits.contiguous()) if logits is not None else None, nT,
                                       ncls, H, W, cellprob_threshold, flow_threshold, niter, min_size,
                                       max_size_fraction, ptr(masks), ptr(cm), ptr(nlab),
                                       ptr(_pp_ws(nT, H, W, dev)), _stream(dev)), "compute_masks")
    return masks, cm, nlab


def masks_to_numpy(masks_i16: torch.Tensor) -> np.ndarray:
    """uint16 instance ids (stored in an int16 tensor) -> numpy uint16."""
    return masks_i16.cpu().numpy().view(np.uint16)


# ---- network building blocks --------------------------------------------------
EPI = dict(bf16=0, gelu=1, resid=2, f32=3, pos=4, relu=5, qkv=6)


def gemm(A: torch.Tensor, Wt: torch.Tensor, epilogue: str = "bf16", bias=None, aux=None):
    M, K = A.shape
    N = Wt.shape[0]
    dev = A.device
    out = torch.empty((M, N), dtype=torch.float32 if epilogue == "f32" else A.dtype, device=dev)
    check(_lib.lib().cpx_gemm(_DT[A.dtype], ptr(A), ptr(Wt), M, N, K, EPI[epilogue], ptr(bias), ptr(aux),
                              ptr(out), N, _stream(dev)), "gemm")
    return out


_DT = {torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16, torch.float32: _lib.DT_F32}


def conv3x3(x: torch.Tensor, Wt: torch.Tensor, epilogue: str = "bf16", bias=None):
    """3x3 / padding-1 convolution over 32 x 32-token images as an implicit GEMM: x (S*1024, C) token-major,
    Wt (N, 9*C) with k = (3 ky + kx) * C + c (``weight.permute(0, 2, 3, 1).reshape(N, 9 * C)`` of a Conv2d)."""
    M, Cc = x.shape
    N = Wt.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(_lib.lib().cpx_conv3x3(_DT[x.dtype], ptr(x), ptr(Wt), M, N, Cc, EPI[epilogue], ptr(bias), ptr(out), N,
                                 _stream(x.device)), "conv3x3")
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6):
    out = torch.empty_like(x)
    check(_lib.lib().cpx_layernorm(_DT[x.dtype], ptr(x), ptr(w), ptr(b), x.shape[0], x.shape[1], eps, ptr(out),
                                   _stream(x.device)), "layernorm")
    return out


def attention(qkv: torch.Tensor, rel_h: torch.Tensor, rel_w: torch.Tensor):
    """qkv (nS*1024, 3072) bf16 / fp16 / fp32; rel_* (64,64) tables of the same type (x8, zero last row)."""
    M = qkv.shape[0]
    nS = M // 1024
    vt = torch.empty((M, 1024), dtype=qkv.dtype, device=qkv.device)
    out = torch.empty((M, 1024), dtype=qkv.dtype, device=qkv.device)
    check(_lib.lib().cpx_attention(_DT[qkv.dtype], ptr(qkv), ptr(rel_h), ptr(rel_w), nS, ptr(vt), ptr(out),
                                   _stream(qkv.device)), "attention")
    return out


def row_stats(x: torch.Tensor) -> torch.Tensor:
    st = torch.empty((x.shape[0], 4, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().cpx_row_stats(ptr(x), x.shape[0], ptr(st), _stream(x.device)), "row_stats")
    return st


def gemm_ln(A, Wt, epilogue="bf16", bias=None, aux=None, ln_stats=None, ln_colsum=None, want_stats=False):
    """GEMM with a LayerNorm over the input row folded in / output row statistics emitted."""
    M, K = A.shape
    N = Wt.shape[0]
    dev = A.device
    out = torch.empty((M, N), dtype=torch.float32 if epilogue == "f32" else A.dtype, device=dev)
    st = torch.zeros((M, 4, 2), dtype=torch.float32, device=dev) if want_stats else None
    check(_lib.lib().cpx_gemm_ln(ptr(A), ptr(Wt), M, N, K, EPI[epilogue], ptr(bias), ptr(aux), ptr(out), N,
                                 ptr(ln_stats), ptr(ln_colsum), ptr(st), _stream(dev)), "gemm_ln")
    return (out, st) if want_stats else out


# ---- f2 ---------------------------------------------------------------------
def dedup_pairs(centers: np.ndarray, max_dist: float = 15 / 2, device=None) -> np.ndarray:
    """``KDTree(centers).query_pairs(max_dist)`` (predict_wsi.py:923-927) as an int32 (P, 2) array of (i, j),
    i < j, sorted by i: the uniform-grid radius search of ``cpx_dedup_pairs`` on the device.
    centers: (n, 2) float64 host array (the rounded centroids)."""
    centers = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 2)
    n = len(centers)
    if n < 2:
        return np.zeros((0, 2), np.int32)
    dev = torch.device(device if device is not None else "cuda")
    cell = 8.0 if max_dist <= 8.0 else float(max_dist)
    lo, hi = centers.min(0), centers.max(0)
    x0, y0 = float(np.floor(lo[0])) - cell, float(np.floor(lo[1])) - cell
    gw, gh = int((hi[0] - x0) // cell) + 2, int((hi[1] - y0) // cell) + 2
    L = _lib.lib()
    c = torch.from_numpy(centers).to(dev)
    nbytes = L.cpx_dedup_pairs_workspace_bytes(n, gw, gh)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tot = torch.zeros(1, dtype=torch.int64, device=dev)
    st = _stream(dev)
    check(L.cpx_dedup_pairs(ptr(c), n, x0, y0, cell, gw, gh, float(max_dist), None, 0, ptr(tot), ptr(ws), nbytes, st),
          "dedup_pairs(count)")
    P = int(tot.item())
    if P == 0:
        return np.zeros((0, 2), np.int32)
    pairs = torch.empty((P, 2), dtype=torch.int32, device=dev)
    check(L.cpx_dedup_pairs(ptr(c), n, x0, y0, cell, gw, gh, float(max_dist), ptr(pairs), P, ptr(tot), ptr(ws), nbytes, st),
          "dedup_pairs(write)")
    return pairs.cpu().numpy()
