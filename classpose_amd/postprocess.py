"""Host half of ``PostProcessor.__call__`` (predict_wsi.py:578-656): id maps + device records
-> polygons with measurements, via the host entry point ``cpx_polygonize_host`` of the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import CpxCell, CpxRecord
from .engine import CELL_DTYPE, RECORD_DTYPE


def polygonize_tile(masks_u16: np.ndarray, records: np.ndarray, scale: float, origin) -> tuple:
    """masks_u16 (H, W) uint16 host array of one tile, records = its RECORD_DTYPE rows.
    Returns (cells CELL_DTYPE[n], xy float64 [n_pts, 2]) in level-0 coordinates."""
    H, W = masks_u16.shape
    n = len(records)
    cells = np.zeros(n, CELL_DTYPE)
    if n == 0:
        return cells, np.zeros((0, 2))
    recs = np.ascontiguousarray(records)
    assert recs.dtype == RECORD_DTYPE and C.sizeof(CpxRecord) == RECORD_DTYPE.itemsize
    masks_u16 = np.ascontiguousarray(masks_u16)
    per = int((recs["y1"] - recs["y0"] + recs["x1"] - recs["x0"]).sum()) * 2 + 8 * n + 64
    while True:
        pool = np.empty((per, 2), np.float64)
        used = _lib.lib().cpx_polygonize_host(masks_u16.ctypes.data, H, W, recs.ctypes.data, n,
                                              float(scale), float(origin[0]), float(origin[1]),
                                              pool.ctypes.data, per, cells.ctypes.data)
        if used == -12:          # CPX_ENOMEM: pool too small (very ragged instances)
            per *= 4
            continue
        if used < 0:
            raise _lib.CpxError(f"cpx_polygonize_host failed with code {used}")
        return cells, pool[:used]
