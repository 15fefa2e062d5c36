"""ctypes binding of libclasspose_hip.so (the C ABI in include/classpose_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a
symbol is absent this module raises, it never routes to another implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclasspose_hip.so")
# the same sources built with -DCPX_DEBUG: the product ABI PLUS include/classpose_hip_debug.h (A/B switches, non-production
# kernel variants, cycle-stamp builds).  Nothing on the product path loads it; tools/*.py and the variant tests do.
DEBUG_LIB_PATH = os.path.join(_HERE, "libclasspose_hip_debug.so")
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 3
DT_BF16, DT_F16, DT_F32 = 0, 1, 2
DTYPE_CODE = {"bf16": DT_BF16, "fp16": DT_F16, "fp32": DT_F32}
PROF_KINDS = ("fc1", "attention", "qkv", "proj", "fc2", "patch_embed", "neck_head")
# the kernel behind kind "fc1" (the dominant launch of the network; its name as rocprofv3 prints it)
FC1_KERNEL_NAME = "k_gemm4w<GELU + folded LayerNorm, one wave per SIMD> = void k_gemm4w<1, 0, false>(Gemm4wArgs)"


class CpxTiling(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("ypad1", C.c_int), ("xpad1", C.c_int),
                ("Ly", C.c_int), ("Lx", C.c_int), ("ny", C.c_int), ("nx", C.c_int),
                ("bsize", C.c_int), ("augment", C.c_int),
                ("ystart", C.c_int * 16), ("xstart", C.c_int * 16)]


class CpxBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "rel_h", "rel_w",
        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "qkv_colsum", "fc1_colsum")]


class CpxConvOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("src_a", C.c_int), ("src_b", C.c_int), ("dst", C.c_int),
                ("cin_a", C.c_int), ("cin_b", C.c_int), ("cout", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("relu", C.c_int), ("weight", C.c_void_p), ("bias", C.c_void_p)]


class CpxQcOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("act", C.c_int),
                ("h_in", C.c_int), ("w_in", C.c_int), ("h_out", C.c_int), ("w_out", C.c_int),
                ("src_a", C.c_size_t), ("src_b", C.c_size_t), ("gate", C.c_size_t), ("res", C.c_size_t),
                ("dst", C.c_size_t),
                ("c_a", C.c_int), ("ld_a", C.c_int), ("up_a", C.c_int), ("c_b", C.c_int), ("ld_b", C.c_int),
                ("ld_res", C.c_int), ("c_out", C.c_int), ("ld_dst", C.c_int), ("c_red", C.c_int),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("w2", C.c_void_p), ("bias2", C.c_void_p)]


class CpxNetWeights(C.Structure):
    _fields_ = [("depth", C.c_int), ("ncls", C.c_int), ("n_head_cols", C.c_int),
                ("ld_head", C.c_int), ("dtype", C.c_int), ("fuse_ln", C.c_int),
                ("pe_w", C.c_void_p), ("pe_b", C.c_void_p), ("pos", C.c_void_p),
                ("blocks", C.POINTER(CpxBlockWeights)),
                ("neck0_w", C.c_void_p), ("neck_ln1_w", C.c_void_p), ("neck_ln1_b", C.c_void_p),
                ("neck2_w", C.c_void_p), ("neck_ln2_w", C.c_void_p), ("neck_ln2_b", C.c_void_p),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p),
                ("n_unet_ops", C.c_int), ("unet_ops", C.POINTER(CpxConvOp)), ("prof", C.c_void_p)]


class CpxWeightJob(C.Structure):
    _fields_ = [("op", C.c_int), ("dtype", C.c_int), ("n", C.c_longlong), ("K", C.c_int), ("reserved", C.c_int),
                ("src_host", C.c_void_p * 4), ("stage_off", C.c_size_t * 4), ("dst", C.c_void_p * 3)]


WJ_ROUND_HALF, WJ_ROUND_F32, WJ_COPY_F32, WJ_FOLD_LN = 0, 1, 2, 3


class CpxRecord(C.Structure):
    _fields_ = [("tile", C.c_int32), ("label", C.c_int32), ("cls", C.c_int32), ("area", C.c_int32),
                ("y0", C.c_int32), ("x0", C.c_int32), ("y1", C.c_int32), ("x1", C.c_int32),
                ("sum_y", C.c_int64), ("sum_x", C.c_int64)]


class CpxCell(C.Structure):
    _fields_ = [("area", C.c_double), ("perimeter", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("n_pts", C.c_int32), ("offset", C.c_int32), ("valid", C.c_int32), ("cls", C.c_int32)]


_p, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/classpose_hip.h declares
SIGNATURES = {
    "cpx_abi_version": (_i, []),
    "cpx_last_error": (C.c_char_p, []),
    "cpx_build_id": (C.c_char_p, []),
    "cpx_normalize_stats_u8": (_i, [_p, _i, _i, _i, _i, _f, _i, _f, _p, _p, _p]),
    "cpx_normalize_apply_u8": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "cpx_resize_linear_u8": (_i, [_p, _i, _i, _i, _p, _i, _i, _p]),
    "cpx_make_subtiles": (_i, [_p, _p, _i, C.POINTER(CpxTiling), _p, _p]),
    "cpx_make_patches": (_i, [_p, _p, _i, C.POINTER(CpxTiling), _i, _p, _p]),
    "cpx_make_subtiles_f32": (_i, [_p, _p, _i, C.POINTER(CpxTiling), _p, _p]),
    "cpx_blend_subtiles": (_i, [_p, _i, _i, _i, C.POINTER(CpxTiling), _p, _p, _p, _p, _p]),
    "cpx_blend_subtiles_nchw": (_i, [_p, _p, _i, _i, C.POINTER(CpxTiling), _p, _p, _p, _p, _p]),
    "cpx_qc_forward": (_i, [C.POINTER(CpxQcOp), _i, _p, _i, _i, _i, _sz, _sz, _i, _i, _p, _p, _p, _sz, _p]),
    "cpx_round_weights": (_i, [_p, _p, C.c_longlong, _i, _i, _p]),
    "cpx_fold_layernorm": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "cpx_weights_build": (_i, [C.POINTER(CpxWeightJob), _i, _p, _sz, _p]),
    "cpx_net_workspace_bytes": (_sz, [_i, _i]),
    "cpx_net_forward": (_i, [C.POINTER(CpxNetWeights), _p, _i, _p, _p, _sz, _p]),
    "cpx_unet_workspace_bytes": (_sz, [C.POINTER(CpxConvOp), _i, _i, _i]),
    "cpx_unet_head_forward": (_i, [C.POINTER(CpxConvOp), _i, _p, _i, _p, _i, _i, _i, _p, _sz, _p]),
    "cpx_net_mlp_parts": (_i, [_i, _i]),
    "cpx_prof_create": (_i, [_i, _i, C.c_uint, C.POINTER(C.c_void_p)]),
    "cpx_prof_collect": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "cpx_prof_collect_launches": (_i, [_p, C.POINTER(C.c_float), C.POINTER(C.c_int), _i, C.POINTER(C.c_int)]),
    "cpx_prof_destroy": (None, [_p]),
    "cpx_gemm": (_i, [_i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p]),
    "cpx_gemm_uses_big_tile": (_i, [_i, _i, _i, _i]),
    "cpx_layernorm": (_i, [_i, _p, _p, _p, _i, _i, _f, _p, _p]),
    "cpx_attention": (_i, [_i, _p, _p, _p, _i, _p, _p, _p]),
    "cpx_gemm_bf16": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p]),
    "cpx_conv3x3": (_i, [_i, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p]),
    "cpx_gemm_ln": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p]),
    "cpx_row_stats": (_i, [_p, _i, _p, _p]),
    "cpx_layernorm_bf16": (_i, [_p, _p, _p, _i, _i, _f, _p, _p]),
    "cpx_attention_relpos": (_i, [_p, _p, _p, _i, _p, _p, _p]),
    "cpx_postproc_workspace_bytes": (_sz, [_i, _i, _i]),
    "cpx_postproc_max_labels": (_i, [_i, _i]),
    "cpx_postproc_launch_count": (C.c_ulonglong, []),
    "cpx_follow_flows": (_i, [_p, _p, _i, _i, _i, _f, _i, _p, _p, _p, _p]),
    "cpx_get_masks": (_i, [_p, _i, _i, _i, _d, _p, _p, _p, _p]),
    "cpx_remove_bad_flow_masks": (_i, [_p, _p, _i, _i, _i, _d, _p, _p, _p]),
    "cpx_fill_holes_and_remove_small_masks": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "cpx_compute_class_masks": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "cpx_remove_border_instances": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "cpx_compute_masks": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _d, _i, _i, _d, _p, _p, _p, _p, _p]),
    "cpx_instance_records": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "cpx_compute_masks_records": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _d, _i, _i, _d, _p, _p, _p, _i, _p, _p, _p, _p]),
    "cpx_find_contours_ccomp_host": (_i, [_p, _i, _i, _p, _i, _p, _p, _p, _i]),
    "cpx_polygonize_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "cpx_polygonize_device": (_i, [_p, _p, _p, _i, _i, _i, _i, _d, _p, _p, _i, _p, _p, _p, _p]),
    "cpx_polygonize_host": (_i, [_p, _i, _i, _p, _i, _d, _d, _d, _p, _i, _p]),
    "cpx_dedup_pairs_workspace_bytes": (_sz, [_i, _i, _i]),
    "cpx_dedup_pairs": (_i, [_p, _i, _d, _d, _d, _i, _i, _d, _p, C.c_longlong, _p, _p, _sz, _p]),
    "cpx_write_geojson": (_i, [C.c_char_p, C.c_char_p, _p, C.c_int64, _p, _p, _p, _p, C.c_int64, _p, _i, _d, _d, _i]),
}
# include/classpose_hip_debug.h: process-global A/B and ablation switches -- exported by libclasspose_hip_debug.so only
_PRIVATE = {
    "cpx_gemm_set_variant": (None, [_i]),
    "cpx_attention_set_trv": (None, [_i]),
    "cpx_attention_set_lsum": (None, [_i]),
    "cpx_attention_set_variant": (None, [_i]),
    "cpx_gemm_set_nt": (None, [_i]),
    "cpx_follow_set_early_exit": (None, [_i]),
    "cpx_follow_set_lds_window": (None, [_i]),
    "cpx_gemm_set_reverse": (None, [_i]),
    "cpx_attention_set_xcd_order": (None, [_i]),
    "cpx_gemm_set_big": (None, [_i]),
    "cpx_gemm_set_persistent": (None, [_i]),
    "cpx_gemm_set_persistent_qkv": (None, [_i]),
    "cpx_attention_debug": (_i, [_p, _p, _p, _i, _p, _p, _p, _p]),
    "cpx_attention8_debug": (_i, [_p, _p, _p, _i, _p, _p, _p, _p]),
    "cpx_attention4_debug": (_i, [_p, _p, _p, _i, _p, _p, _p, _p]),
    "cpx_attention2q_debug": (_i, [_p, _p, _p, _p, _i, _p, _p, _p]),
    "cpx_attention2q_set_ablation": (None, [_i]),
    "cpx_postproc_set_fused": (None, [_i]),
    "cpx_gemm_set_split": (None, [_i]),
    "cpx_gemm_set_direct": (None, [_i]),
    "cpx_gemm4w": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "cpx_gemm4w_set_variant": (None, [_i]),
    "cpx_gemm_set_4w": (None, [_i]),
    "cpx_net_set_mlp_parts": (None, [_i]),
    "cpx_gemm_set_balanced": (None, [_i]),
    "cpx_gemm_set_dbg": (None, [_i]),
    "cpx_gemm_set_l2_block": (None, [_i]),
    "cpx_gemm_set_pingpong": (None, [_i]),
    "cpx_gemm_set_epi4": (None, [_i]),
    "cpx_gemm_pingpong_occupancy": (_i, []),
    "cpx_gemm_pingpong_stamps": (_i, [_p, _sz]),
    "cpx_gemm_set_pingpong_opts": (None, [_i, _i]),
}

_lib = None


class CpxError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into classpose_amd/libclasspose_hip.so."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


def source_build_id() -> str:
    """What ``cpx_build_id()`` of a library built from the sources on disk returns (csrc/Makefile: BUILD_ID)."""
    import glob
    import hashlib
    names = sorted(os.path.basename(f) for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(CSRC, pat)))
    names = sorted(names + ["Makefile"])
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(_HERE), "include")
    for f in [os.path.join(CSRC, n) for n in names] + [os.path.join(inc, "classpose_hip.h"), os.path.join(inc, "classpose_hip_debug.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_BUILD_ID_CHECKED: set = set()


def build_id() -> str:
    """``cpx_build_id()`` of the loaded library; a mismatch with the sources on disk is reported once on stderr (the .so
    files are git-ignored and travel with snapshots: this is how a stale one shows)."""
    bid = lib().cpx_build_id().decode()
    if bid in _BUILD_ID_CHECKED:
        return bid
    _BUILD_ID_CHECKED.add(bid)
    try:
        src = source_build_id()
    except OSError:
        return bid
    if bid.split("+")[0] != src:
        import sys
        print(f"classpose_amd: {bid} was not built from the sources on disk ({src}): run `make -C classpose_amd/csrc`", file=sys.stderr)
    return bid


def _load(path: str, signatures: dict):
    if not os.path.exists(path):
        raise CpxError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: it ships its own libamdhip64 and the host side hands us ITS streams and device pointers.  Loaded before torch, this library would
    # bind the system HIP runtime instead, and a process would hold two runtimes -- ours without a device context ("no ROCm-capable device is
    # detected" at the first launch; seen when build() and smoke() ran in one interpreter on a GPU box, round 5)
    import torch  # noqa: F401
    L = C.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(L, name)          # AttributeError if the symbol is absent -> loud
        fn.restype = res
        fn.argtypes = args
    if L.cpx_abi_version() != ABI_VERSION:
        raise CpxError(f"{os.path.basename(path)} ABI version mismatch")
    return L


def lib():
    """The library every product call goes through: libclasspose_hip.so, unless this process has switched to the debug
    build (``use_debug_library()`` or CLASSPOSE_HIP_DEBUG=1 in the environment: tools/*.py, variant tests)."""
    global _lib
    if _lib is None:
        if os.environ.get("CLASSPOSE_HIP_DEBUG") == "1":
            _lib = _load(DEBUG_LIB_PATH, {**SIGNATURES, **_PRIVATE})
        else:
            _lib = _load(LIB_PATH, SIGNATURES)
    return _lib


_debug_lib = None


class use_debug_library:
    """``with _lib.use_debug_library() as L:`` -- inside the block ``lib()`` (hence ``ops.*``) is the -DCPX_DEBUG build,
    whose switches ``L.cpx_*_set_*`` select kernel variants; objects created outside the block (an ``Engine``) keep the
    product library."""

    def __enter__(self):
        global _lib, _debug_lib
        if _debug_lib is None:
            _debug_lib = _load(DEBUG_LIB_PATH, {**SIGNATURES, **_PRIVATE})
        self._saved = _lib
        _lib = _debug_lib
        return _debug_lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().cpx_last_error()
        raise CpxError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor / numpy array; None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
