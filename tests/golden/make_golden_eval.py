"""Mint a golden vector from the reference's OWN per-tile orchestration (build container only):

* ``ClassposeModel.eval``          /root/reference/src/classpose/models.py:478-827 (list branch + 2-D branch)
* ``ClassposeModel._run_net``      models.py:347-426
* ``ClassposeModel._compute_masks`` / ``compute_masks``   models.py:97-188, 428-476
* ``core.run_net`` / ``_forward``  core.py:51-231, ``compute_class_masks`` models.py:191-230

called exactly as the WSI worker calls them (predict_wsi.py:751-757: ``model.eval([tile], batch_size=.., augment=tta,
bsize=bsize, compute_masks=True)``) on a fake ``self`` that carries an elementwise stand-in network.  What lives in the
ABSENT cellpose wheel is supplied at the call boundary by the oracle's restatements (``transforms.normalize_img``,
tiling helpers, ``dynamics.resize_and_compute_masks``), so those stay "unpinned"; what the vector PINS is everything the
reference owns in between: which functions run in which order with which arguments (normalisation parameters, bsize /
augment / tile_overlap / batch_size plumbing, the channel split and transposes of ``_run_net``, niter = 200 and the
thresholds handed to the dynamics, min_size, the class vote on the final masks, the squeeze / return structure).  The
keyword arguments the reference passes across the boundary are recorded next to the outputs.

The input tile is regenerated from a seed by the test; fixtures hold data only.   python tests/golden/make_golden_eval.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402

NCLS = 4


def eval_tile(seed: int, size: int = 256) -> np.ndarray:
    """uint8 RGB tile whose channels ENCODE cell-like fields of the procedural nuclei (dY, dX around 128, inside / outside
    as 200 / 50): after the per-channel 1-99 percentile normalisation the elementwise network below decodes them again, so
    the dynamics downstream find real cells."""
    from classpose_amd import synth
    dP, cp, _, _ = synth.analytic_fields(seed, 64, 32, size, size, NCLS)
    img = np.stack([128.0 + 20.0 * dP[0], 128.0 + 20.0 * dP[1], np.where(cp > 0, 200.0, 50.0)], -1)
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(img + rng.integers(-2, 3, img.shape)), 0, 255).astype(np.uint8)


def decode_net(X: torch.Tensor, ncls: int) -> torch.Tensor:
    """(B, 3, b, b) normalised image -> (B, ncls + 3, b, b) [class logits | dY | dX | cellprob]; only +, -, * on float32"""
    dy = (X[:, 0] - 0.5) * 12.0
    dx = (X[:, 1] - 0.5) * 12.0
    cp = (X[:, 2] - 0.5) * 12.0
    cls = [X[:, 2] * 0.0 + 1.0 - X[:, 2] * 2.0]                      # background wins outside
    for c in range(1, ncls):
        cls.append(X[:, 0] * (0.5 * c) + X[:, 1] * (1.5 - 0.5 * c) + X[:, 2] - 1.0)
    return torch.stack(cls + [dy, dx, cp], 1)


def main():
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, mg.REF)
    from oracle import dynamics as odyn
    from oracle import tiling
    from classpose_amd import models as our_models
    import cellpose.transforms as ctf
    ctf.get_pad_yx, ctf.make_tiles = tiling.get_pad_yx, tiling.make_tiles
    ctf.average_tiles, ctf.unaugment_tiles = tiling.average_tiles, tiling.unaugment_tiles
    import cellpose.core as ccore
    ccore.tqdm_out = None
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm")).trange = range
    import classpose.models as rm

    log = {}

    def convert_image(x, channel_axis=None, z_axis=None, do_3D=False):
        log["convert_image"] = dict(channel_axis=channel_axis, z_axis=z_axis, do_3D=bool(do_3D), shape=list(x.shape), dtype=str(x.dtype))
        return x

    def normalize_img(x, **params):
        log["normalize_img"] = {k: (v if isinstance(v, (int, float, bool, str, type(None))) else repr(v)) for k, v in params.items()}
        return tiling.normalize_img(x)

    def resize_image(x, Ly=None, Lx=None, no_channels=False, interpolation=None):
        shp = x.shape[:2] if no_channels and x.ndim == 2 else (x.shape[-3:-1] if not no_channels else x.shape[-2:])
        assert tuple(shp) == (Ly, Lx), f"resize_image stand-in: {x.shape} -> {(Ly, Lx)} is not the identity"
        log.setdefault("resize_image_calls", 0)
        log["resize_image_calls"] += 1
        return x

    def resize_and_compute_masks(dP, cellprob, **kw):
        log["resize_and_compute_masks"] = {k: (v if isinstance(v, (int, float, bool, str, type(None))) else repr(v)) for k, v in kw.items()}
        log["resize_and_compute_masks"]["dP_shape"] = list(dP.shape)
        assert kw.get("resize") is None
        return odyn.compute_masks(np.ascontiguousarray(dP), np.ascontiguousarray(cellprob), niter=kw["niter"],
                                  cellprob_threshold=kw["cellprob_threshold"], flow_threshold=kw["flow_threshold"],
                                  min_size=kw["min_size"], max_size_fraction=kw["max_size_fraction"])

    rm.transforms.convert_image = convert_image
    rm.transforms.normalize_img = normalize_img
    rm.transforms.resize_image = resize_image
    rm.dynamics.resize_and_compute_masks = resize_and_compute_masks
    rm.plot.dx_to_circ = our_models.dx_to_circ
    rm.utils.TqdmToLogger = lambda *a, **k: None             # progress-bar plumbing of the absent wheel
    # cellpose.models.normalize_default as published in cellpose 4.0.8 (a constant of the absent wheel)
    rm.normalize_default = {"lowhigh": None, "percentile": None, "normalize": True, "norm3D": True, "sharpen_radius": 0,
                            "smooth_radius": 0, "tile_norm_blocksize": 0, "tile_norm_smooth3D": 1, "invert": False}

    class FakeNet(nn.Module):
        def __init__(self, ncls):
            super().__init__()
            self.n_cell_classes, self.device = ncls, torch.device("cpu")
            self.dummy = nn.Parameter(torch.zeros(1))

        def forward(self, X):
            return decode_net(X, self.n_cell_classes), torch.zeros(X.shape[0], 256)

    me = types.SimpleNamespace(net=FakeNet(NCLS), nclasses=NCLS, device=torch.device("cpu"), timing=[])
    for name in ("eval", "_run_net", "_compute_masks", "_resize_cellprob", "_resize_gradients"):
        if hasattr(rm.ClassposeModel, name):
            setattr(me, name, types.MethodType(getattr(rm.ClassposeModel, name), me))

    out = {}
    for k, (seed, tta, bs) in enumerate(((31, False, 8), (32, True, 4))):
        tile = eval_tile(seed)
        log.clear()
        with torch.no_grad():
            masks, flows, class_masks, styles = me.eval([tile], batch_size=bs, augment=tta, bsize=256, compute_masks=True)
        assert isinstance(masks, list) and len(masks) == 1 and len(class_masks) == 1
        out[f"ev_{k}_cfg"] = np.array([seed, int(tta), bs])
        out[f"ev_{k}_tilesum"] = np.array(int(tile.astype(np.int64).sum()))
        out[f"ev_{k}_masks"] = np.asarray(masks[0]).astype(np.uint16)
        out[f"ev_{k}_class_masks"] = np.asarray(class_masks[0]).astype(np.uint8)
        out[f"ev_{k}_dP"] = np.asarray(flows[0][1])[:, ::4, ::4].astype(np.float32)
        out[f"ev_{k}_cellprob"] = np.asarray(flows[0][2])[::4, ::4].astype(np.float32)
        out[f"ev_{k}_yclass"] = np.asarray(flows[0][3])[:, ::4, ::4].astype(np.float32)
        out[f"ev_{k}_shape_x"] = np.array(flows[0][4])
        out[f"ev_{k}_n_flows"] = np.array(len(flows[0]))
        import json
        out[f"ev_{k}_log"] = np.array(json.dumps(log, sort_keys=True))
        print(k, "cells:", int(np.asarray(masks[0]).max()), "classes:", np.unique(class_masks[0]).tolist(), json.dumps(log, sort_keys=True)[:400])
    out["ev_n"] = np.array(2)
    np.savez_compressed(os.path.join(HERE, "reference_eval.npz"), **out)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "reference_eval.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
