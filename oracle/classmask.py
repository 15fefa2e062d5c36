"""Oracle (test infrastructure): class vote, border removal, dedup, records.

CPU restatement of reference functions that ARE importable under the stub
importer, so every function here is PINNED by golden vectors minted from the
reference itself (tests/golden/make_golden.py):

* ``compute_class_masks``       /root/reference/src/classpose/models.py:191-230
* ``remove_border_instances``   /root/reference/src/classpose/metrics/pq.py:65-92
* ``deduplicate``               /root/reference/src/classpose/entrypoints/predict_wsi.py:896-965

plus the per-instance records the device hands to the host in place of the
pickled (masks, class_masks) arrays of predict_wsi.py:757-763 (our own
contract; checked against numpy here).

Not imported by anything under ``classpose_amd/``.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import KDTree


def compute_class_masks(masks: np.ndarray, y_class: np.ndarray):
    """models.py:191-230.  masks (H,W) int, y_class (ncls,[1,]H,W) float."""
    sq = y_class.squeeze()
    cls_px = sq.argmax(axis=0)
    inst = masks.ravel()
    cls = cls_px.ravel()
    max_inst = int(inst.max())
    n_classes = int(sq.shape[0])
    valid = inst > 0
    idx = inst[valid].astype(np.int64) * n_classes + cls[valid]
    counts = np.bincount(idx, minlength=(max_inst + 1) * n_classes)
    counts = counts.reshape(max_inst + 1, n_classes)
    major = counts.argmax(axis=1)
    major[0] = 0
    class_masks = major[masks]
    return class_masks, np.unique(masks)


def remove_border_instances(mask: np.ndarray) -> np.ndarray:
    """metrics/pq.py:65-92 (in place, returns the same array)."""
    instances = mask[..., 0] if mask.ndim == 3 else mask
    border = np.unique(np.concatenate(
        [instances[0], instances[:, 0], instances[-1], instances[:, -1]]))
    border = border[border != 0]
    mask[np.isin(instances, border)] = 0
    return mask


def deduplicate_indices(centers: np.ndarray, sizes: np.ndarray, max_dist: float = 15 / 2):
    """predict_wsi.py:896-965 on bare arrays; returns indices kept.

    The greedy grouping iterates a Python ``set`` of index pairs exactly like
    the reference (``tree.query_pairs`` returns a set; iteration order is the
    set's), so results are identical for identical inputs in the same
    interpreter build.
    """
    tree = KDTree(centers)
    neighbours = tree.query_pairs(max_dist)
    groups: dict[int, list] = {}
    member_to_group: dict[int, int] = {}
    for pair in neighbours:
        if (pair[0] not in member_to_group) and (pair[1] not in member_to_group):
            group_idx = len(groups)
            groups[group_idx] = []
            member_to_group[pair[0]] = group_idx
            member_to_group[pair[1]] = group_idx
        else:
            if pair[0] in member_to_group:
                group_idx = member_to_group[pair[0]]
            else:
                group_idx = member_to_group[pair[1]]
        if pair[0] not in groups[group_idx]:
            groups[group_idx].append(pair[0])
        if pair[1] not in groups[group_idx]:
            groups[group_idx].append(pair[1])
    to_remove = {}
    for k in groups:
        group = groups[k]
        if len(group) > 1:
            curr_sizes = [sizes[i] for i in group]
            largest = group[np.argmax(curr_sizes)]
            for i in group:
                if i != largest and i not in to_remove:
                    to_remove[i] = True
    return [i for i in range(len(centers)) if i not in to_remove]


def instance_records(masks: np.ndarray, class_masks: np.ndarray):
    """Per-label pixel statistics: the compact record our device path emits.

    Returns dict of arrays indexed by label-1: area (int64), bbox y0,x0,y1,x1
    (inclusive-exclusive, int32), sum_y, sum_x (int64), cls (int32 = class of the
    first pixel in raster order, predict_wsi.py:634).
    """
    K = int(masks.max())
    area = np.zeros(K, np.int64)
    bbox = np.zeros((K, 4), np.int32)
    sy = np.zeros(K, np.int64)
    sx = np.zeros(K, np.int64)
    cls = np.zeros(K, np.int32)
    for l in range(1, K + 1):
        yy, xx = np.nonzero(masks == l)
        if yy.size == 0:
            continue
        area[l - 1] = yy.size
        bbox[l - 1] = [yy.min(), xx.min(), yy.max() + 1, xx.max() + 1]
        sy[l - 1] = yy.sum()
        sx[l - 1] = xx.sum()
        cls[l - 1] = class_masks[yy[0], xx[0]]
    return dict(area=area, bbox=bbox, sum_y=sy, sum_x=sx, cls=cls)
