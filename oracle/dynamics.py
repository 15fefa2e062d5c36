"""Oracle (test infrastructure): flows -> instance masks, on CPU.

Restatement of cellpose==4.0.8 (pinned in /root/reference/uv.lock:352-353; the
package is NOT vendored under /root/reference and is absent from this image):

* ``dynamics.resize_and_compute_masks`` / ``compute_masks``   <- models.py:149-159
* ``dynamics.follow_flows`` / ``steps_interp``                 
* ``dynamics.get_masks_torch`` / ``max_pool_nd``               
* ``dynamics.remove_bad_flow_masks``, ``metrics.flow_error``,
  ``dynamics.masks_to_flows_gpu`` / ``_extend_centers_gpu``    
* ``utils.fill_holes_and_remove_small_masks``                  
* fastremap==1.17.7 ``unique / mask / renumber`` semantics, fill-voids==2.1.1
  ``fill`` (== ``scipy.ndimage.binary_fill_holes`` default structure).

PARITY UNPINNED: the reference holds no golden vector for any of this
(tests/test_prediction_integration.py:76-82 asserts file existence only).
Where cellpose itself calls a library that is in this image
(``torch.nn.functional.grid_sample``, ``scipy.ndimage.find_objects/mean``)
the oracle calls it too.  Deliberate, documented choices where the pinned
implementation is order-undefined:

* seed ordering: ``argsort`` of counts is made STABLE (ties keep raster order);
* the 9-neighbour mean of the fp64 heat diffusion is summed in index order
  0..8 and divided by 9 (torch's reduction order for a [9, N] mean depends on
  N and on the thread count; differences are 1 ulp of fp64);
* label quirk KEPT: ``fill_holes_and_remove_small_masks`` indexes
  ``unique(...)[1][1:]`` by position, so after ``remove_bad_flow_masks`` has
  left gaps in the label set the first size filter removes label ``i+1`` where
  ``i`` is the POSITION of the small label among the present labels.

Not imported by anything under ``classpose_amd/``.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy import ndimage


# --------------------------------------------------------------------------
# fastremap stand-ins (semantics only)
# --------------------------------------------------------------------------
def fr_unique(a: np.ndarray):
    return np.unique(a, return_counts=True)


def fr_mask(a: np.ndarray, labels) -> np.ndarray:
    out = a.copy()
    labels = np.asarray(labels)
    if labels.size:
        out[np.isin(out, labels)] = 0
    return out


def fr_renumber(a: np.ndarray) -> np.ndarray:
    """fastremap.renumber(preserve_zero=True): 1..K in C-order first appearance."""
    flat = a.ravel()
    uniq, first = np.unique(flat, return_index=True)
    order = np.argsort(first, kind="stable")
    lut = {}
    k = 1
    for u in uniq[order]:
        if u == 0:
            continue
        lut[int(u)] = k
        k += 1
    out = np.zeros_like(flat)
    for u, v in lut.items():
        out[flat == u] = v
    return out.reshape(a.shape)


# --------------------------------------------------------------------------
# a11: follow_flows (literal torch path of cellpose.dynamics.steps_interp)
# --------------------------------------------------------------------------
def steps_interp(dP: np.ndarray, inds, niter: int) -> torch.Tensor:
    """dP (2,H,W) float32 already masked and /5; inds = (y_idx, x_idx).
    Returns float32 tensor (2, N): rows [y, x]."""
    shape = dP.shape[1:]
    ndim = len(shape)
    n = len(inds[0])
    pt = torch.zeros((1, 1, n, ndim), dtype=torch.float32)
    im = torch.zeros((1, ndim, *shape), dtype=torch.float32)
    for k in range(ndim):
        pt[0, 0, :, ndim - k - 1] = torch.from_numpy(np.asarray(inds[k])).to(torch.float32)
        im[0, ndim - k - 1] = torch.from_numpy(np.ascontiguousarray(dP[k])).to(torch.float32)
    shp = np.array(shape)[::-1].astype("float") - 1
    for k in range(ndim):
        im[:, k] *= 2.0 / shp[k]
        pt[..., k] /= shp[k]
    pt *= 2
    pt -= 1
    for _ in range(niter):
        dPt = torch.nn.functional.grid_sample(im, pt, align_corners=False)
        for k in range(ndim):
            pt[..., k] = torch.clamp(pt[..., k] + dPt[:, k], -1.0, 1.0)
    pt += 1
    pt *= 0.5
    for k in range(ndim):
        pt[..., k] *= shp[k]
    pt = pt[..., [1, 0]].squeeze()
    if pt.ndim == 1:          # single point: squeeze dropped the N axis
        pt = pt[None]
    return pt.T


def follow_flows(dP: np.ndarray, inds, niter: int = 200) -> torch.Tensor:
    return steps_interp(dP, inds, niter)


# --------------------------------------------------------------------------
# a12: get_masks_torch
# --------------------------------------------------------------------------
def _max_pool1d(h: torch.Tensor, kernel_size: int, axis: int) -> torch.Tensor:
    out = h.clone()
    nd = h.shape[axis]
    k0 = kernel_size // 2
    for d in range(-k0, k0 + 1):
        if axis == 1:
            mv = out[:, max(-d, 0):min(nd - d, nd)]
            hv = h[:, max(d, 0):min(nd + d, nd)]
        else:
            mv = out[:, :, max(-d, 0):min(nd - d, nd)]
            hv = h[:, :, max(d, 0):min(nd + d, nd)]
        torch.maximum(mv, hv, out=mv)
    return out


def max_pool_nd(h: torch.Tensor, kernel_size: int = 5) -> torch.Tensor:
    hmax = _max_pool1d(h, kernel_size, 1)
    return _max_pool1d(hmax, kernel_size, 2)


def get_masks(pt: torch.Tensor, inds, shape0, rpad: int = 20,
              max_size_fraction: float = 0.4, return_debug: bool = False):
    """pt int tensor (2, N) final positions; inds (y_idx, x_idx)."""
    ndim = len(shape0)
    pt = pt.clone().long()
    pt += rpad
    pt = torch.clamp(pt, min=0)
    for i in range(len(pt)):
        pt[i] = torch.clamp(pt[i], max=shape0[i] + rpad - 1)
    shape = tuple(np.array(shape0) + 2 * rpad)
    h1 = torch.zeros(shape, dtype=torch.int32)
    h1.index_put_(tuple(pt), torch.ones(pt.shape[1], dtype=torch.int32), accumulate=True)
    hmax1 = max_pool_nd(h1.unsqueeze(0), kernel_size=5).squeeze(0)
    seeds1 = torch.nonzero((h1 - hmax1 > -1e-6) * (h1 > 10))
    if len(seeds1) == 0:
        z = np.zeros(shape0, dtype="uint16")
        return (z, dict(h1=h1.numpy(), seeds=np.zeros((0, 2), np.int64))) if return_debug else z
    npts = h1[tuple(seeds1.T)]
    isort1 = torch.from_numpy(np.argsort(npts.numpy(), kind="stable"))
    seeds1 = seeds1[isort1]
    n_seeds = len(seeds1)
    h_slc = torch.zeros((n_seeds, *[11] * ndim))
    for k in range(n_seeds):
        slc = tuple(slice(int(seeds1[k][j]) - 5, int(seeds1[k][j]) + 6) for j in range(ndim))
        h_slc[k] = h1[slc]
    seed_masks = torch.zeros((n_seeds, *[11] * ndim))
    seed_masks[:, 5, 5] = 1
    for _ in range(5):
        seed_masks = max_pool_nd(seed_masks, kernel_size=3)
        seed_masks *= h_slc > 2
    M1 = torch.zeros(shape, dtype=torch.int64)
    for k in range(n_seeds):
        nz = torch.nonzero(seed_masks[k]) + seeds1[k] - 5
        M1[tuple(nz.T)] = 1 + k
    M1p = M1[tuple(pt)].numpy()
    dtype = "uint16" if n_seeds < 2 ** 16 else "uint32"
    M0 = np.zeros(shape0, dtype=dtype)
    M0[inds] = M1p
    uniq, counts = fr_unique(M0)
    big = np.prod(shape0) * max_size_fraction
    bigc = uniq[counts > big]
    if len(bigc) > 0 and (len(bigc) > 1 or bigc[0] != 0):
        M0 = fr_mask(M0, bigc)
    M0 = fr_renumber(M0).reshape(tuple(shape0))
    if return_debug:
        return M0, dict(h1=h1.numpy(), seeds=seeds1.numpy(), M1=M1.numpy())
    return M0


# --------------------------------------------------------------------------
# a13: flow-error filter
# --------------------------------------------------------------------------
_NBR_DY = [0, -1, 1, 0, 0, -1, -1, 1, 1]
_NBR_DX = [0, 0, 0, -1, 1, -1, 1, -1, 1]


def get_centers(masks: np.ndarray, slices):
    """cellpose.dynamics.get_centers: in-mask pixel closest to the centroid,
    ext = bbox_h + bbox_w + 2."""
    centers = np.zeros((len(slices), 2), "int32")
    ext = np.zeros((len(slices),), "int32")
    for p, si in enumerate(slices):
        i, y0, y1, x0, x1 = si
        yi, xi = np.nonzero(masks[y0:y1, x0:x1] == (i + 1))
        ymed = yi.mean()
        xmed = xi.mean()
        imin = ((xi - xmed) ** 2 + (yi - ymed) ** 2).argmin()
        centers[p] = [yi[imin] + y0, xi[imin] + x0]
        ext[p] = (y1 - y0) + (x1 - x0) + 2
    return centers, ext


def masks_to_flows(masks: np.ndarray, niter: int | None = None, return_debug=False):
    """masks_to_flows_gpu + _extend_centers_gpu, fp64, returns (2,H,W) float64."""
    Ly0, Lx0 = masks.shape
    if masks.max() == 0:
        return np.zeros((2, Ly0, Lx0), "float32")
    mp = np.pad(masks.astype(np.int64), 1)
    y, x = np.nonzero(mp)
    nb_y = np.stack([y + d for d in _NBR_DY])
    nb_x = np.stack([x + d for d in _NBR_DX])
    m0 = mp[nb_y[0], nb_x[0]]
    isneighbor = np.ones((9, y.shape[0]), bool)
    for i in range(1, 9):
        isneighbor[i] = mp[nb_y[i], nb_x[i]] == m0
    sl = ndimage.find_objects(masks)
    slices = [(i, s[0].start, s[0].stop, s[1].start, s[1].stop)
              for i, s in enumerate(sl) if s is not None]
    centers, ext = get_centers(masks, slices)
    meds = centers.astype(np.int64) + 1
    n_iter = int(2 * ext.max()) if niter is None else niter
    T = np.zeros(mp.shape, np.float64)
    isn = isneighbor.astype(np.float64)
    for _ in range(n_iter):
        # ``T[tuple(meds.T)] += 1`` is an advanced-index assignment: duplicate
        # centre coordinates add 1 once (they cannot repeat: one centre per label)
        T[meds[:, 0], meds[:, 1]] += 1
        Tn = T[nb_y, nb_x] * isn
        s = Tn[0].copy()
        for i in range(1, 9):
            s = s + Tn[i]
        T[nb_y[0], nb_x[0]] = s / 9
    dy = T[nb_y[2], nb_x[2]] - T[nb_y[1], nb_x[1]]
    dx = T[nb_y[4], nb_x[4]] - T[nb_y[3], nb_x[3]]
    mu = np.stack((dy, dx)).astype("float64")
    mu /= 1e-60 + (mu ** 2).sum(axis=0) ** 0.5
    mu0 = np.zeros((2, Ly0, Lx0))
    mu0[:, y - 1, x - 1] = mu
    if return_debug:
        return mu0, dict(centers=centers, ext=ext, n_iter=n_iter, T=T)
    return mu0


def flow_error(maski: np.ndarray, dP_net: np.ndarray):
    dP_masks = masks_to_flows(maski)
    flow_errors = np.zeros(maski.max())
    for i in range(dP_masks.shape[0]):
        flow_errors += ndimage.mean((dP_masks[i] - dP_net[i] / 5.0) ** 2, maski,
                                    index=np.arange(1, maski.max() + 1))
    return flow_errors, dP_masks


def remove_bad_flow_masks(masks: np.ndarray, flows: np.ndarray, threshold: float = 0.4,
                          return_errors=False):
    merrors, _ = flow_error(masks, flows)
    badi = 1 + (merrors > threshold).nonzero()[0]
    masks = masks.copy()
    masks[np.isin(masks, badi)] = 0
    return (masks, merrors) if return_errors else masks


# --------------------------------------------------------------------------
# a14: hole fill + size filter
# --------------------------------------------------------------------------
def fill_holes_and_remove_small_masks(masks: np.ndarray, min_size: int = 15) -> np.ndarray:
    masks = masks.copy()
    if min_size > 0:
        counts = fr_unique(masks)[1][1:]
        masks = fr_mask(masks, np.nonzero(counts < min_size)[0] + 1)
        masks = fr_renumber(masks)
    slices = ndimage.find_objects(masks)
    j = 0
    for i, slc in enumerate(slices):
        if slc is not None:
            msk = masks[slc] == (i + 1)
            msk = ndimage.binary_fill_holes(msk)      # == fill_voids.fill in 2-D
            masks[slc][msk] = j + 1
            j += 1
    if min_size > 0:
        counts = fr_unique(masks)[1][1:]
        masks = fr_mask(masks, np.nonzero(counts < min_size)[0] + 1)
        masks = fr_renumber(masks)
    return masks


# --------------------------------------------------------------------------
# a11-a14 driver: dynamics.resize_and_compute_masks (resize=None)
# --------------------------------------------------------------------------
def compute_masks(dP: np.ndarray, cellprob: np.ndarray, niter: int = 200,
                  cellprob_threshold: float = 0.0, flow_threshold: float = 0.4,
                  min_size: int = 15, max_size_fraction: float = 0.4,
                  return_stages: bool = False):
    """dP (2,H,W) float32, cellprob (H,W) float32 -> uint16 (H,W) instance ids."""
    stages = {}
    if (cellprob > cellprob_threshold).sum():
        inds = np.nonzero(cellprob > cellprob_threshold)
        p_final = follow_flows(dP * (cellprob > cellprob_threshold) / 5.0, inds, niter)
        stages["p_final"] = p_final.numpy().copy()
        p_int = p_final.int()
        mask = get_masks(p_int, inds, dP.shape[1:], max_size_fraction=max_size_fraction)
        stages["masks_seeded"] = mask.copy()
        if mask.max() > 0 and flow_threshold is not None and flow_threshold > 0:
            mask, errs = remove_bad_flow_masks(mask, dP, threshold=flow_threshold,
                                               return_errors=True)
            stages["flow_errors"] = errs
        stages["masks_flowfiltered"] = mask.copy()
        if mask.max() < 2 ** 16 and mask.dtype != "uint16":
            mask = mask.astype("uint16")
    else:
        mask = np.zeros(cellprob.shape, "uint16")
        return (mask, stages) if return_stages else mask
    mask = fill_holes_and_remove_small_masks(mask, min_size=min_size)
    return (mask, stages) if return_stages else mask
