"""Oracle (test infrastructure): ctypes loader for oracle/c/dynamics_ref.c."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "dynamics_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_follow_flows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_void_p]
        _lib.oracle_follow_flows.restype = None
    return _lib


def follow_flows(dP: np.ndarray, inds, niter: int = 200) -> np.ndarray:
    """Same contract as oracle.dynamics.follow_flows, returns float32 (2, N)."""
    dP = np.ascontiguousarray(dP, np.float32)
    iy = np.ascontiguousarray(inds[0], np.int32)
    ix = np.ascontiguousarray(inds[1], np.int32)
    out = np.empty((2, iy.size), np.float32)
    lib().oracle_follow_flows(dP.ctypes.data, dP.shape[1], dP.shape[2], iy.ctypes.data,
                              ix.ctypes.data, iy.size, niter, out.ctypes.data)
    return out
