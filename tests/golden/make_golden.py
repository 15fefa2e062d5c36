"""Mint golden vectors from the REFERENCE ITSELF (run in the build container only).

/root/reference is pure Python but its third-party wheels (cellpose,
segment_anything, cv2, openslide, shapely, skimage, matplotlib colormaps, ...) are
absent here, so the reference's modules are imported under a ``sys.meta_path``
stub finder that fabricates empty modules for the missing top-level packages.
Only reference functions whose bodies are pure numpy / torch / scipy are then
CALLED; their inputs and outputs are frozen into ``tests/golden/*.npz|json``.
Nothing from /root/reference is copied: fixtures hold data only.

    python tests/golden/make_golden.py      # rewrites tests/golden/
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"
STUBBED = ("cellpose", "segment_anything", "cv2", "openslide", "shapely", "skimage",
           "fastremap", "fill_voids", "numba", "torchvision", "timm",
           "segmentation_models_pytorch", "h5py", "geopandas", "spatialdata",
           "matplotlib", "huggingface_hub", "pylibCZIrw", "anndata", "dask", "xarray_schema",
           "PIL")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name in ("Transformer", "CellposeModel"):
            import torch
            cls = type(name, (torch.nn.Module,), {})
        elif name == "colormaps":
            class _CM(dict):
                def __getitem__(self, k):
                    return types.SimpleNamespace(colors=[(0.5, 0.5, 0.5)] * 12)
            return _CM()
        else:
            cls = type(name, (), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUBBED:
            try:      # real package present? use it
                if fullname.split(".")[0] in ("matplotlib", "PIL"):
                    for f in sys.meta_path:
                        if f is self:
                            continue
                        spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                        if spec is not None:
                            return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    from classpose.models import compute_class_masks
    from classpose.metrics.pq import remove_border_instances
    from classpose.transforms.transforms import unaugment_class_tiles
    from classpose.unet import UNet
    from classpose.entrypoints import predict_wsi as pw
    import torch

    rng = np.random.default_rng(20260101)
    out = {}

    # ---- remove_border_instances: the reference's 9 known-answer cases + random
    def inst6():
        m = np.zeros((6, 6), np.int64)
        m[0:3, 0:3] = 1; m[0:3, 3:6] = 2; m[2:4, 2:4] = 3; m[3:6, 3:6] = 4
        return m
    cases = [inst6()]
    m = np.zeros((4, 4), np.int64); m[0:2] = 1; m[2:4] = 2; cases.append(m)
    cases.append(np.zeros((5, 5), np.int64))
    m = np.zeros((5, 5), np.int64); m[1:4, 1:4] = 7; cases.append(m)
    i6 = inst6(); c6 = np.zeros_like(i6)
    for a, b in ((1, 1), (2, 2), (3, 3), (4, 1)):
        c6[i6 == a] = b
    cases.append(np.stack([i6, c6], -1))
    m = np.zeros((4, 4), np.int64); m[0:2] = 1; m[2:4] = 2
    cases.append(np.stack([m, np.ones_like(m)], -1))
    for k in range(6):   # random blobs incl. ragged shapes
        H, W = int(rng.integers(3, 70)), int(rng.integers(3, 70))
        m = np.zeros((H, W), np.int64)
        for lab in range(1, int(rng.integers(2, 12))):
            y, x = rng.integers(0, H), rng.integers(0, W)
            m[y:y + rng.integers(1, 9), x:x + rng.integers(1, 9)] = lab
        cases.append(m if k % 2 == 0 else np.stack([m, (m * 3) % 5], -1))
    for i, cse in enumerate(cases):
        out[f"rbi_in_{i}"] = cse.copy()
        out[f"rbi_out_{i}"] = remove_border_instances(cse.copy())
    out["rbi_n"] = np.array(len(cases))

    # ---- compute_class_masks: random + ties + class-0 majority + empty
    ccm = []
    for k in range(8):
        H, W, ncls = int(rng.integers(8, 64)), int(rng.integers(8, 64)), int(rng.integers(2, 11))
        masks = np.zeros((H, W), np.uint16)
        for lab in range(1, int(rng.integers(1, 20))):
            y, x = rng.integers(0, H), rng.integers(0, W)
            masks[y:y + rng.integers(1, 12), x:x + rng.integers(1, 12)] = lab
        if k == 0:
            masks[:] = 0
        y_class = rng.standard_normal((ncls, 1, H, W)).astype(np.float32)
        if k in (1, 2):     # quantised logits => many exact ties
            y_class = np.round(y_class).astype(np.float32)
        if k == 3:          # class 0 wins everywhere
            y_class[0] += 10
        ccm.append((masks, y_class))
    for i, (masks, y_class) in enumerate(ccm):
        cm, uniq = compute_class_masks(masks, y_class)
        out[f"ccm_masks_{i}"] = masks; out[f"ccm_logits_{i}"] = y_class
        out[f"ccm_out_{i}"] = cm; out[f"ccm_uniq_{i}"] = uniq
    out["ccm_n"] = np.array(len(ccm))

    # ---- unaugment_class_tiles on a 3x3 grid
    y = rng.standard_normal((3, 3, 4, 6, 5)).astype(np.float32)
    out["uct_in"] = y.copy()
    out["uct_out"] = unaugment_class_tiles(y.copy())

    # ---- UNet head forward, seeded weights (same topology as fts=[64,128], small widths
    #      so the fixture stays small: 2 encoder levels + bottleneck, 32x32 input)
    torch.manual_seed(7)
    un = UNet(in_channels=16, out_channels=24, n_channels=[8, 12]).eval()
    xin = torch.randn(2, 16, 32, 32)
    with torch.no_grad():
        yout = un(xin)
    out["unet_x"] = xin.numpy()
    out["unet_y"] = yout.numpy()
    for k, v in un.state_dict().items():
        out["unet_sd__" + k] = v.numpy()

    # ---- the same with FOUR encoder levels (1 x 1 at the bottom of the bottleneck) and ragged widths
    torch.manual_seed(8)
    un4 = UNet(in_channels=16, out_channels=24, n_channels=[6, 10, 12, 20]).eval()
    xin4 = torch.randn(1, 16, 32, 32)
    with torch.no_grad():
        yout4 = un4(xin4)
    out["unet4_x"] = xin4.numpy()
    out["unet4_y"] = yout4.numpy()
    for k, v in un4.state_dict().items():
        out["unet4_sd__" + k] = v.numpy()

    # ---- _get_coords (bound method is a plain generator; call unbound with a dummy self)
    gc = {}
    for dim, tile, ov in ((10000, 256, 32), (40000, 256, 32), (80000, 256, 32),
                          (40000, 512, 32), (1000, 256, 64), (300, 256, 32), (255, 256, 0)):
        coords = list(pw.SlideLoader._get_coords(None, tile, ov, (dim, dim), 1.0))
        gc[f"{dim}_{tile}_{ov}"] = dict(n=len(coords), first=coords[:3], last=coords[-3:])
    coords = list(pw.SlideLoader._get_coords(None, 256, 32, (1000, 700), 2.0))
    gc["rect_1000x700_ts2"] = dict(n=len(coords), all=coords)

    # ---- deduplicate on synthetic clusters (chain A-B-C, pairs, singletons)
    def feat(cx, cy, area, i):
        cell = dict(id=str(i), coords=[[cx, cy], [cx + 1, cy], [cx + 1, cy + 1], [cx, cy]],
                    label="a", color=[1, 2, 3], area=area, perimeter=4.0, centroid=[cx, cy])
        return pw.to_geojson_polygon(cell)
    pts = []
    for k in range(300):
        pts.append((float(np.round(rng.random() * 400, 2)), float(np.round(rng.random() * 400, 2)),
                    float(rng.integers(20, 400))))
    pts += [(500.0, 500.0, 50.0), (505.0, 500.0, 60.0), (510.0, 500.0, 55.0),    # chain
            (600.0, 600.0, 10.0), (600.0, 607.4, 10.0), (600.0, 607.6 + 7.5, 99.0)]
    feats = [feat(cx, cy, a, i) for i, (cx, cy, a) in enumerate(pts)]
    kept = pw.deduplicate(feats)
    gj = dict(points=pts, kept_ids=[int(f["id"]) for f in kept],
              feature0=feats[0], centroid0=pw.polygons_to_centroids(feats[:1])[0])
    gj["centroid0"]["id"] = "uuid"
    shifted = pw.apply_bounds_offset_to_feature(json.loads(json.dumps(feats[1])), 10.0, 20.0)
    gj["feature1"] = feats[1]
    gj["feature1_shifted"] = shifted

    np.savez_compressed(os.path.join(HERE, "reference_functions.npz"), **out)
    with open(os.path.join(HERE, "reference_functions.json"), "w") as f:
        json.dump(dict(get_coords=gc, geojson=gj), f, indent=1)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
