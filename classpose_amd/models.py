"""``ClassposeModel`` -- the reference's in-process model API on the MI355X engine.

Mirrors /root/reference/src/classpose/models.py:233-827 for the 2-D path used by the WSI
worker (predict_wsi.py:716-756) and by paper_experiments/run_inference.py:196-208: same
constructor arguments, same ``eval`` signature and return tuple
``(masks uint16, (rgb_flow, dP, cellprob, y_class, x.shape), class_masks int64, styles)``.
Everything between the uint8 image and the id maps runs on the GPU through the C ABI; the
3-D / stitching / diameter-rescale branches of the reference are outside the WSI path and
raise NotImplementedError instead of silently doing something else.
"""
from __future__ import annotations

import threading

import numpy as np
import torch

from . import engine, ops
from .log import get_logger

models_logger = get_logger(__name__)
PRECISIONS = ("fp32", "fp16", "bf16")


def resolve_precision(precision: str, device: torch.device | None = None) -> str:
    """models.py:37-69.  gfx950 has bf16 MFMA, so bf16 never falls back to fp16; fp32 runs on the
    exact-f32 matrix instruction (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) with float32
    weights and activations end to end."""
    if precision not in PRECISIONS:
        raise ValueError(f"Unknown precision '{precision}'. Expected one of {sorted(PRECISIONS)}.")
    models_logger.info("Using inference precision: %s", precision)
    return precision


def dx_to_circ(dP: np.ndarray) -> np.ndarray:
    """cellpose.plot.dx_to_circ (flow visualisation returned as flows[0])."""
    mag = np.sqrt(np.sum(dP ** 2, axis=0)).astype(np.float32)
    x01, x99 = np.percentile(mag, 1), np.percentile(mag, 99)
    mag = (mag - x01) / (x99 - x01) if x99 - x01 > 1e-3 else np.zeros_like(mag)
    mag = 255 * np.clip(mag, 0, 1.0) / 2
    ang = np.arctan2(dP[1], dP[0]) + np.pi
    rgb = np.zeros((*dP.shape[1:], 3), "uint8")
    for k in range(3):
        rgb[..., k] = np.clip(mag * (np.cos(ang + 2 * np.pi * k / 3) + 1), 0, 255).astype("uint8")
    return rgb


class ClassposeModel:
    def __init__(self, gpu: bool = True, pretrained_model: str | dict = "cpsam", model_type=None,
                 diam_mean=None, device: torch.device | None = None, nchan=None, nclasses: int = 3,
                 feature_transformation_structure: list[int] | None = None, precision: str = "bf16",
                 max_batch_tiles: int = 8, max_engines: int = 4):
        if device is None:
            device = torch.device("cuda:0")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the MI355X engine has no CPU path: pass a cuda device")
        self.device = device
        self.precision = resolve_precision(precision, device)
        sd = pretrained_model if isinstance(pretrained_model, dict) else \
            torch.load(pretrained_model, map_location="cpu", weights_only=True)
        self.weights = engine.NetWeights.from_state_dict(sd, self.precision, device)
        if nclasses not in (None, self.weights.ncls):
            models_logger.warning("nclasses=%s but the checkpoint has %s classes; using the checkpoint",
                                  nclasses, self.weights.ncls)
        self.nclasses = self.weights.ncls
        self.max_batch_tiles = max_batch_tiles
        # several Python threads may share one model and call eval concurrently (the reference's --inference_threads,
        # predict_wsi.py:728-798): the weights are shared and read-only; an ENGINE (network + post-processing work spaces, output
        # slots, two HIP streams: ~0.9 GB for 8 tiles of 256 px at ViT-L, growing with the tile area) is checked out of a bounded pool
        # per (tile shape, options) for the duration of one batch and returned afterwards.  At most `max_engines` exist per key, so N
        # concurrent callers overlap on the GPU up to that number and queue beyond it, and a caller that starts a fresh thread per
        # request re-uses the pool's engines instead of leaking one per thread (round 4 keyed the table by thread id and never pruned).
        self.max_engines = max(1, int(max_engines))
        self._pool: dict = {}                    # key -> {"idle": [Engine], "n": engines created}
        self._cv = threading.Condition()
        self.timing = []

    class _Lease:
        def __init__(self, model, key, make):
            self.model, self.key, self.make, self.eng = model, key, make, None

        def __enter__(self):
            m = self.model
            with m._cv:
                slot = m._pool.setdefault(self.key, {"idle": [], "n": 0})
                while True:
                    if slot["idle"]:
                        self.eng = slot["idle"].pop()
                        return self.eng
                    if slot["n"] < m.max_engines:
                        slot["n"] += 1                       # reserve; built outside the lock (allocations take tens of ms)
                        break
                    m._cv.wait()
            try:
                self.eng = self.make()
            except BaseException:
                with m._cv:
                    slot["n"] -= 1
                    m._cv.notify()
                raise
            return self.eng

        def __exit__(self, *exc):
            m = self.model
            with m._cv:
                m._pool[self.key]["idle"].append(self.eng)
                m._cv.notify()

    def _engine(self, H, W, augment, tile_overlap, kw):
        """``with self._engine(...) as eng:`` -- an engine of the bounded pool for this tile shape and these options"""
        key = (H, W, bool(augment), float(tile_overlap), tuple(sorted(kw.items())))
        return self._Lease(self, key, lambda: engine.Engine(self.weights, H, W, batch_tiles=self.max_batch_tiles,
                                                            augment=augment, tile_overlap=tile_overlap, **kw))

    def engines_alive(self) -> int:
        with self._cv:
            return sum(slot["n"] for slot in self._pool.values())

    def eval(self, x, batch_size: int = 8, resample: bool = True, channels=None, channel_axis=None,
             z_axis=None, normalize=True, invert: bool = False, rescale=None, diameter=None,
             flow_threshold: float = 0.4, cellprob_threshold: float = 0.0, do_3D: bool = False,
             anisotropy=None, flow3D_smooth=0, stitch_threshold: float = 0.0, min_size: int = 15,
             max_size_fraction: float = 0.4, niter=None, augment: bool = False,
             tile_overlap: float = 0.1, bsize: int = 256, compute_masks: bool = True, progress=None):
        if do_3D or stitch_threshold > 0 or diameter is not None or invert or normalize is not True:
            raise NotImplementedError("only the 2-D WSI-tile path of ClassposeModel.eval is built "
                                      "(no do_3D / stitch / diameter / invert / custom normalize)")
        if bsize != 256:
            raise NotImplementedError("bsize is fixed to 256 like the WSI worker (predict_wsi.py:1563)")
        is_list = isinstance(x, (list, tuple))
        imgs = list(x) if is_list else [x]
        outs = [None] * len(imgs)
        # images of equal shape are batched across the list (the reference loops one by one)
        groups: dict = {}
        for i, im in enumerate(imgs):
            im = np.asarray(im)
            if im.ndim != 3 or im.shape[-1] != 3 or im.dtype != np.uint8:
                raise ValueError("expected uint8 (H, W, 3) RGB tiles")
            groups.setdefault(im.shape, []).append(i)
        kw = dict(niter=200 if not niter else int(niter), cellprob_threshold=cellprob_threshold,
                  flow_threshold=flow_threshold, min_size=min_size, max_size_fraction=max_size_fraction)
        nT = self.max_batch_tiles
        for shape, idxs in groups.items():
            for s in range(0, len(idxs), nT):
                chunk = idxs[s:s + nT]
                tiles = torch.from_numpy(np.stack([np.ascontiguousarray(imgs[i]) for i in chunk])).to(self.device)
                with self._engine(shape[0], shape[1], augment, tile_overlap, kw) as eng:     # held for this batch only
                    o = eng.run(tiles, records=False)
                    masks = ops.masks_to_numpy(o.masks) if compute_masks else None
                    cm = o.class_masks.cpu().numpy().astype(np.int64) if compute_masks else None
                    dP, cp = o.dP.cpu().numpy(), o.cellprob.cpu().numpy()
                    yc = o.logits.cpu().numpy() if o.logits is not None else None
                for k, i in enumerate(chunk):
                    outs[i] = (masks[k].copy() if compute_masks else np.zeros(0),
                               (dx_to_circ(dP[k]), dP[k].copy(), cp[k].copy(),
                                yc[k].copy() if yc is not None else None, (1, *shape)),
                               cm[k].copy() if compute_masks else np.zeros(0),
                               np.zeros(256, np.float32))          # styles: noise in the reference (vit_sam.py:197)
        if is_list:
            return ([o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], [o[3] for o in outs])
        return outs[0]
