"""The tile loop behind ``classpose-predict-wsi`` (everything that needs torch / the HIP library).

``entrypoints/predict_wsi.py`` is the light front -- flags, the parent that spawns one worker per listed GPU -- and forwards
to ``run_main`` here; it also re-exports every public name of this module (``TileStream``, ``run_rank``, ``CELL_ROW`` ...), so
callers keep importing from ``classpose_amd.entrypoints.predict_wsi``.  Split in round 6: the parent of a multi-GPU run used to
import torch (1 - 1.6 s) only to spawn children that import it again.

Reference: /root/reference/src/classpose/entrypoints/predict_wsi.py (``main`` :1451-1886, ``worker`` :659-810,
``SlideLoader`` :183-451, ``PostProcessor`` :578-656, ``deduplicate`` :896-965).

  slide -> tile grid (``_get_coords``) -> [per GPU] tiles streamed through pinned memory
  -> Engine (normalise, sub-tile, ClassTransformer, blend, dynamics, class vote, records)
  -> id maps + records to the host -> polygons/measurements (``cpx_polygonize_host``)
  -> all-gather of the cell tables across ranks (RCCL) -> de-duplication -> GeoJSON.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import queue
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from .. import engine, geojson, hooks, hostinfo, ops, parallel, postprocess, roi, wsi
from ..log import get_logger
from ..model_configs import DEFAULT_MODEL_CONFIGS, ModelConfig

from .predict_wsi import (DEFAULT_OVERLAP, DEFAULT_TILE_SIZE, GEOJSON_OUTPUT_TEMPLATES, MIN_TILE_SIZE,  # noqa: F401
                          _args_dict, _check_unsupported, get_geojson_output_filename)

logger = get_logger("classpose")

def get_device(device: str | None = None) -> list[torch.device]:
    """utils.get_device: ``cuda:0,1,2`` -> one device per index."""
    if device is not None:
        if ":" in device:
            kind, idxs = device.split(":")
            return [torch.device(f"{kind}:{i}") for i in idxs.split(",")]
        return [torch.device(device)]
    if torch.cuda.is_available():
        return [torch.device("cuda")]
    raise RuntimeError("no GPU visible: the MI355X engine has no CPU path")


CELL_ROW = np.dtype([("area", "<f8"), ("perimeter", "<f8"), ("cx", "<f8"), ("cy", "<f8"),
                     ("n_pts", "<i8"), ("cls", "<i8")])


class TileStream:
    """Reader side of the tile loop: slide regions are decoded by a pool of threads (OpenSlide and the
    procedural slide release the GIL) several batches ahead, land in pinned host batches and go to the
    device with hipMemcpyAsync on a side stream, so reads / copies overlap the engine.  ``extra``
    (optional) computes per-tile side inputs in the same pool (flow-injection tests)."""

    def __init__(self, slide, plan, idxs, nT, H, W, device, depth: int = 3, extra=None, autostart: bool = True,
                 gate_at: int | None = None):
        self.slide, self.plan, self.idxs, self.nT = slide, plan, list(idxs), nT
        self.dev, self.extra = device, extra
        # optional gate: batch gate_at is copied to the device but not handed over, later batches are neither copied nor handed
        # over before release() -- bench.py times a region that starts with the reader no further than its steady state: ONE batch
        # resident on the device (the steady state has up to `depth` of them queued), up to `ahead` decoded into pinned host
        # memory.  `parked` is set once the reader waits at the gate with that copy issued.
        self.gate_at, self.gate, self.parked = gate_at, threading.Event(), threading.Event()
        self._stop = threading.Event()
        self._pe_lock = threading.Lock()
        # (small tiles are many short numpy / decoder calls that take turns at the GIL: 32 threads still beat 12 on a 16-core cgroup quota -- the
        # 40 000^2 slide at 256 px read at 360 against 221 tiles/s -- so the quota only caps the pool at twice its cores)
        n_workers = max(2, min(32, (os.cpu_count() or 4) // 2, hostinfo.usable_cpus() * 2))
        self.ahead = max(depth, -(-n_workers // max(nT, 1)))        # batches being decoded at once
        self.q: queue.Queue = queue.Queue(maxsize=depth)
        self.pinned = [torch.empty((nT, H, W, 3), dtype=torch.uint8).pin_memory()
                       for _ in range(self.ahead + depth + 2)]
        self.copied: dict = {}
        self.pinned_extra: dict = {}             # slot -> tuple of pinned [nT, ...] float arrays (allocated when the first side input arrives)
        self.copy_stream = torch.cuda.Stream(device)
        self.readers = ThreadPoolExecutor(max_workers=n_workers)
        self.t = threading.Thread(target=self._run, daemon=True)
        self._started = False
        if autostart:
            self.start()

    def release(self):
        self.gate.set()

    def close(self):
        """stop reading: wakes a reader parked at the gate (a consumer that failed before ``release()`` would otherwise
        leave the thread, its pool and the pinned buffers waiting until process exit) and lets ``_run`` wind down"""
        self._stop.set()
        self.gate.set()
        while True:                                  # unblock a producer parked on the full queue
            try:
                self.q.get_nowait()
            except queue.Empty:
                break

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def start(self):
        """begin reading / copying (``autostart=False``: buffers and threads exist, nothing has been read yet)"""
        if not self._started:
            self._started = True
            self.t.start()

    def _submit(self, b):
        chunk = self.idxs[b * self.nT:(b + 1) * self.nT]
        slot = b % len(self.pinned)
        if slot in self.copied:                                  # its previous H2D copy must have left the buffer
            self.copied.pop(slot).synchronize()
        host = self.pinned[slot]

        def read(k, ti):
            host[k].copy_(torch.from_numpy(wsi.read_tile(self.slide, self.plan, self.plan.coords[ti])))
            if self.extra is None:
                return None
            # side inputs of a plug-in (flow-injection tests): every reader thread writes its tile's arrays straight into the slot's pinned
            # batch -- until round 4 the CONSUMER np.stack-ed the batch (320 MB for 8 tiles of 1024 px) and copied it from pageable memory
            # on its own thread, 180 ms per batch against 140 ms of GPU work
            arrs = self.extra(ti)
            pe = self.pinned_extra.get(slot)
            if pe is None:
                with self._pe_lock:
                    pe = self.pinned_extra.get(slot)
                    if pe is None:
                        # (a side input may be None -- the class logits of a model without a class head: no buffer, None on the device side too)
                        pe = tuple(None if a is None else
                                   torch.empty((self.nT,) + tuple(a.shape), dtype=torch.from_numpy(np.asarray(a)).dtype).pin_memory() for a in arrs)
                        self.pinned_extra[slot] = pe
            for dst, a in zip(pe, arrs):
                if dst is not None:
                    dst[k].copy_(torch.from_numpy(np.ascontiguousarray(a)))
            return True
        return chunk, slot, [self.readers.submit(read, k, ti) for k, ti in enumerate(chunk)]

    def _run(self):
        try:
            n_batches = -(-len(self.idxs) // self.nT)
            pending = []
            nxt = 0
            while (nxt < n_batches or pending) and not self._stop.is_set():
                while nxt < n_batches and len(pending) < self.ahead:
                    pending.append((nxt,) + self._submit(nxt))       # decode into pinned host memory, up to `ahead` batches ahead
                    nxt += 1
                if self._stop.is_set() or not pending:
                    break
                b, chunk, slot, futs = pending.pop(0)
                extras = [f.result() for f in futs]
                gated = self.gate_at is not None and b >= self.gate_at and not self.gate.is_set()
                if gated and b > self.gate_at:
                    # the gate holds back the H2D copy and the hand-over of these batches, not their decoding
                    while not self.gate.wait(0.5):
                        pass
                    if self._stop.is_set():
                        break
                with torch.cuda.stream(self.copy_stream):
                    dev = self.pinned[slot][: len(chunk)].to(self.dev, non_blocking=True)
                    if extras and extras[0] is not None:           # the side inputs of the batch, device tensors by the same event
                        extras = tuple(None if t is None else t[: len(chunk)].to(self.dev, non_blocking=True) for t in self.pinned_extra[slot])
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                self.copied[slot] = ev
                if gated and b == self.gate_at:
                    # the first gated batch waits here, resident on the device: a step of the steady state finds its tiles there too
                    # (they were copied while the step before it ran) -- until round 5 this copy and the reader's wake-up sat at the
                    # head of bench.py's timed region with the GPU idle, ~3 ms of a 20-step run
                    self.parked.set()
                    while not self.gate.wait(0.5):
                        pass
                    if self._stop.is_set():
                        break
                while not self._stop.is_set():
                    try:
                        self.q.put((chunk, dev, ev, extras), timeout=0.5)
                        break
                    except queue.Full:
                        pass
            if not self._stop.is_set():
                self.q.put(None)
        except BaseException as e:      # surface reader errors in the consumer
            self.q.put(e)
        finally:
            self.readers.shutdown(wait=False)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item


def run_rank(args, rank: int, world: int, device: torch.device):
    """Everything one GPU does; returns (cell table, vertex pool) of this rank's shard."""
    hooks.load_plugins()
    if getattr(args, "model_path", None) is not None:
        # predict_wsi_cpsam: a plain Cellpose-SAM checkpoint, no semantic head, every cell is "cell"
        model_config = ModelConfig(path=args.model_path, url=None, mpp=args.train_mpp, cell_types=[])
    elif args.model_config in DEFAULT_MODEL_CONFIGS:
        model_config = ModelConfig(**DEFAULT_MODEL_CONFIGS[args.model_config])
    else:
        model_config = ModelConfig.load_from_yaml(args.model_config)
    model_config.download_if_necessary()
    torch.cuda.set_device(device)
    # wall-clock attribution of this rank (main thread), logged at the end: where a run's seconds go besides the GPU
    stage = {k: 0.0 for k in ("checkpoint load", "slide open + plan", "tissue detection", "wait for weights", "engine + stream setup",
                              "reader wait", "resize + submit", "device wait", "polygons D2H + cell rows")}
    t_s = time.time()
    sd = model_config.load_state_dict()
    fts, n_classes, _ = engine.NetWeights.infer_structure(sd)
    stage["checkpoint load"] = time.time() - t_s
    # the 0.6 GB of network weights are converted and uploaded on a background thread while this one opens the slide, plans the grid, runs
    # the tissue detection and starts the readers (until round 4 this sat serially in front of the first tile: ~1.5 s of a ~9 s start-up)
    bg = ThreadPoolExecutor(max_workers=2, thread_name_prefix="cpx-bg")

    def _make_weights():
        torch.cuda.set_device(device)
        t_w = time.time()
        w_ = engine.NetWeights.from_state_dict(sd, args.precision, device)
        return w_, time.time() - t_w
    weights_fut = bg.submit(_make_weights)
    t_s = time.time()
    if getattr(args, "model_path", None) is not None:
        if n_classes > 1:
            raise ValueError("--model_path expects a Cellpose-SAM checkpoint without a class head; "
                             "use classpose-predict-wsi --model_config for Classpose models")
        labels = None
    elif model_config.cell_types:
        if len(model_config.cell_types) != n_classes - 1:
            raise ValueError(f"Number of labels ({len(model_config.cell_types)}) does not match "
                             f"number of classes ({n_classes - 1})")
        labels = model_config.cell_types
    else:
        labels = [str(i) for i in range(1, n_classes)]
    slide = wsi.WSIReader(args.slide_path)
    plan = wsi.plan_slide(slide, args.tile_size, args.overlap, model_config.mpp)
    if rank == 0:
        logger.info(f"Slide MPP: {plan.mpp}; model MPP: {model_config.mpp}; tiles: {len(plan.coords)}; "
                    f"slide dimensions: {plan.slide_dim}; tile {args.tile_size}/{args.overlap}")
    rois = None
    if args.roi_geojson:
        # ROI mode (predict_wsi.py:233-236,251-256,441-446): ROI-driven grid, tiles that miss
        # every ROI are skipped; ROI tiles may be smaller than --tile_size (one engine per size)
        need_classes = bool(args.output_type) and "csv" in args.output_type
        loaded = roi.load_roi_polygons(args.roi_geojson, group_by_class=need_classes)
        if loaded is None:
            raise ValueError(f"no polygons in {args.roi_geojson}")
        rois, plan.roi_class_dict = loaded if need_classes else (loaded, None)
        if plan.bounds != (0.0, 0.0):
            rois = [g.translate(*plan.bounds) for g in rois]
        coords = roi.get_coords_roi(rois, plan.read_tile_size, plan.read_overlap, args.overlap, plan.ts)
        plan.coords = [c for c in coords if roi.check_tile_in_cnts(c[0], c[1], plan.ts, rois)]
        if rank == 0:
            logger.info(f"Selecting tiles using ROI with {len(rois)} polygons: {len(plan.coords)} tiles")
    plan.rois = rois
    plan.tissue_cnts = None
    stage["slide open + plan"] = time.time() - t_s
    t_s = time.time()
    if args.tissue_detection_model_path is not None:
        # SlideLoader._get_tissue_contours (predict_wsi.py:305-322): GrandQC tissue detection on the
        # 10 um/px thumbnail before the first tile is read; tiles that miss every tissue polygon are skipped
        from .. import grandqc
        logger.info("Detecting tissue contours using GrandQC")
        _, _, _, tissue_cnts, _, _ = grandqc.detect_tissue_wsi(
            wsi.WSIReader(args.slide_path), model_td_path=args.tissue_detection_model_path,
            min_area=args.min_area, device=device, class_map_override=_qc_override("tissue"))
        plan.tissue_cnts = roi.polygons_from_cnts(tissue_cnts)
        logger.info(f"Number of tissue contours: {len(plan.tissue_cnts)}")
        if not plan.tissue_cnts:
            logger.warning("No tissue detected in slide. Skipping inference.")
            plan.coords = []
        else:
            plan.coords = [c for c in plan.coords if roi.check_tile_in_cnts(c[0], c[1], plan.ts, plan.tissue_cnts)]
        if rank == 0:
            logger.info(f"Tiles to predict after the tissue filter: {len(plan.coords)}")
    stage["tissue detection"] = time.time() - t_s
    # GrandQC artefact detection (predict_wsi.py:1678-1760 of the reference runs it after the tile loop): its result is needed by the final
    # cell filter only, so rank 0 starts it NOW on a background thread -- thumbnail rendering / JPEG round trip on the host, ~300 patches on
    # the GPU beside the tile loop -- and write_outputs collects it (8 s of a 41 s run were spent serially behind the loop in round 4)
    plan.artefact_future = None
    if rank == 0 and args.artefact_detection_model_path is not None and args.tissue_detection_model_path is not None and plan.coords:
        plan.artefact_future = bg.submit(_detect_artefacts, args, device)
    weights = None
    mine = list(parallel.shard_indices(len(plan.coords), rank, world))
    by_size: dict[int, list[int]] = {}
    for ti in mine:
        by_size.setdefault(plan.coords[ti][1], []).append(ti)
    pool = ThreadPoolExecutor(max_workers=max(2, min(16, hostinfo.usable_cpus() // max(world, 1))))
    futures = []
    cells_all, xy_all, tile_all, n_invalid = [], [], [], 0
    scale = plan.polygon_scale        # min(train_mpp / mpp_x, train_mpp / mpp_y) from the shared slots, predict_wsi.py:1517-1524
    t0 = time.time()
    n_done = 0
    for R, idxs in by_size.items():                              # R: pixels read per tile side
        H, W = ops.resized_shape(R, R, plan.resize_factor)       # what the network sees
        n_sub = engine.make_tiling(H, W, 256, args.tta).ny ** 2
        # sub-tiles per launch: >= 96 (+7..9 % over 32), and a multiple of 8 so that the 256-row GEMM grids keep their
        # 8 x 4 super-tile order (tools/bench_variants.py: 9-sub-tile geometries 10 -> 16 tiles per launch +3..5 %,
        # the 25-sub-tile one 3 -> 8 tiles +4.5 %); --batch_size can raise it
        nT = max(1, max(args.batch_size, 96) // n_sub)
        step = 8 // math.gcd(n_sub, 8)
        nT = -(-nT // step) * step
        t_s = time.time()
        extra = None
        provider = hooks.field_provider(slide, plan, n_classes) if hooks.field_provider else None
        if provider is not None:                                 # plug-in supplied dynamics inputs (classpose_amd/hooks.py)
            def extra(ti, R=R, W=W, H=H):
                return provider(ti, R, W, H)[:3]                 # (dP, cellprob, logits)
        # The readers start decoding now -- unless the checkpoint is still on its way to the device AND this rank has few cores to itself:
        # the upload's host side is one thread copying 1.2 GB of mapped pages into the runtime's staging buffers, and with eight ranks on a
        # 16-core quota the readers' threads starved it (0.3 s of work took 1.6 - 3.0 s; nothing can run before the weights are there anyway)
        weights_first = weights is None and not weights_fut.done() and hostinfo.usable_cpus() // max(world, 1) < 8
        stream = TileStream(slide, plan, idxs, nT, R, R, device, extra=extra, autostart=not weights_first)
        waited = 0.0
        if weights is None:
            t_w = time.time()
            weights, t_conv = weights_fut.result()
            waited = stage["wait for weights"] = time.time() - t_w
            logger.info(f"[rank {rank}] weights converted + uploaded in {t_conv:.2f} s on a background thread "
                        f"(this thread waited {waited:.2f} s for them{'; the readers start now' if weights_first else ''})")
        stream.start()
        eng = engine.Engine(weights, H, W, batch_tiles=nT, augment=args.tta)
        stage["engine + stream setup"] += time.time() - t_s - waited

        def collect(sid, chunk, keep_alive):
            nonlocal n_done, n_invalid
            n = len(chunk)
            t_c = time.time()
            out = eng.result(sid)
            if int(out.nlabels.max()) >= 65535:              # (the first read-back of the batch: this is where the host waits for the GPU)
                raise RuntimeError("more than 65535 instances in one tile: uint16 ids would wrap")
            stage["device wait"] += time.time() - t_c
            t_c = time.time()
            polys = eng.fetch_polygons(n, out)
            if polys is not None:                                    # contours traced on the device (f1)
                cells, tile_in_batch, xy = polys
                keep = cells["valid"] == 1
                n_invalid += int((~keep).sum())
                xy_all.append(geojson.gather_vertices(xy, cells["offset"][keep], cells["n_pts"][keep]))
                rows = np.zeros(int(keep.sum()), CELL_ROW)
                for name in ("area", "perimeter", "cx", "cy", "n_pts", "cls"):
                    rows[name] = cells[keep][name]
                cells_all.append(rows)
                tile_all.append(np.asarray(chunk, np.int64)[tile_in_batch[keep]])       # global tile index of every cell
            else:                                                    # vertex pool overflow: host polygoniser
                masks = out.masks.cpu().numpy().view(np.uint16)      # D2H: 2 B / pixel
                recs = eng.fetch_records(n, out)
                for k, ti in enumerate(chunk):
                    origin = plan.coords[ti][0]
                    futures.append((ti, pool.submit(postprocess.polygonize_tile, masks[k].copy(),
                                                    recs[recs["tile"] == k], scale, origin)))
            stage["polygons D2H + cell rows"] += time.time() - t_c
            n_done += n
            if rank == 0 and (n_done // nT) % 20 == 0:
                logger.info(f"Predicted tiles: {n_done}/{len(mine)} "
                            f"({n_done / max(time.time() - t0, 1e-9):.1f} tiles/s/GPU)")

        in_flight = None                     # one batch runs on the device while the previous one is collected
        it = iter(stream)
        while True:
            t_r = time.time()
            try:
                chunk, tiles_dev, ev, f = next(it)
            except StopIteration:
                break
            stage["reader wait"] += time.time() - t_r
            t_r = time.time()
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            # the raw batch was allocated on the copy stream: tell the allocator that this stream reads it too,
            # or its block could be handed to the next H2D copy while the resize kernel is still reading
            tiles_dev.record_stream(cur)
            tiles_dev = ops.resize_tile_to_target_mpp(tiles_dev, plan.resize_factor)
            inject = None
            if extra is not None:                # the plug-in's fields arrive as device tensors of the same copy-stream event (TileStream)
                inject = tuple(f)
                for t_ in inject:
                    if t_ is not None:
                        t_.record_stream(cur)
            sid = eng.submit(tiles_dev, inject=inject, records=True,
                             polygons=(scale, [plan.coords[ti][0] for ti in chunk]))
            stage["resize + submit"] += time.time() - t_r
            if in_flight is not None:
                collect(*in_flight)
            in_flight = (sid, chunk, (tiles_dev, inject))
        if in_flight is not None:
            collect(*in_flight)
        del eng, stream
    for ti, f in futures:
        cells, xy = f.result()
        keep = cells["valid"] == 1
        n_invalid += int((~keep).sum())
        xy_all.append(geojson.gather_vertices(xy, cells["offset"][keep], cells["n_pts"][keep]))
        rows = np.zeros(int(keep.sum()), CELL_ROW)
        for name in ("area", "perimeter", "cx", "cy", "n_pts", "cls"):
            rows[name] = cells[keep][name]
        cells_all.append(rows)
        tile_all.append(np.full(len(rows), ti, np.int64))
    pool.shutdown()
    slide.close()
    cells = np.concatenate(cells_all) if cells_all else np.zeros(0, CELL_ROW)
    xy = np.concatenate(xy_all) if xy_all else np.zeros((0, 2))
    plan.cell_tiles = np.concatenate(tile_all) if tile_all else np.zeros(0, np.int64)
    logger.info(f"[rank {rank}] {len(cells)} cells, {n_invalid} invalid, {len(mine)} tiles in {time.time() - t0:.1f}s")
    # main-thread wall seconds per stage ("device wait" = blocked on the GPU's results, i.e. the engine is the pace-maker; "reader wait" = blocked
    # on decoded tiles [+ the plug-in's fields]; the rest is host work between two batches)
    logger.info(f"[rank {rank}] stage wall (s): " + ", ".join(f"{k} {v:.2f}" for k, v in stage.items()))
    plan.stage_wall = stage
    bg.shutdown(wait=False)
    return cells, xy, labels, plan


def canonical_cell_order(cells: np.ndarray, xy: np.ndarray, tiles: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Cells in the order of their tile's index in the ``_get_coords`` walk (within a tile: label order), whatever rank,
    batch, tile-size group or fallback path produced them.  The reference's order is whatever its worker processes happen
    to deliver (a shared queue); de-duplication depends on it inside clusters of >= 3 cells, so THIS order is what makes the
    output of an N-GPU run identical to the 1-GPU run (static sharding)."""
    if len(cells) == 0 or bool(np.all(tiles[1:] >= tiles[:-1])):
        return cells, xy
    order = np.argsort(tiles, kind="stable")
    offs = np.concatenate([[0], np.cumsum(cells["n_pts"])])[:-1]
    return cells[order], geojson.gather_vertices(xy, offs[order], cells["n_pts"][order])


def gather_cells(cells: np.ndarray, xy: np.ndarray, device, tiles: np.ndarray | None = None):
    """The path's one exchange (north_star: "RCCL only to all-gather the per-tile centroid / class tensors"):
    the 48-byte cell rows (centroid, class, area, perimeter, vertex count) and the cells' tile indices go to EVERY rank
    (one padded all-gather each); the polygon vertex pools -- ~0.5 KB per cell, needed by the writing rank only -- go to
    rank 0 by point-to-point sends and never leave the device on the other ranks.  Returns (cells, xy, tiles); xy is None
    on ranks other than 0."""
    c = torch.from_numpy(cells.view(np.uint8).reshape(len(cells), CELL_ROW.itemsize).copy()).to(device)
    c = parallel.all_gather_records(c).cpu().numpy()
    out_cells = c.reshape(-1).view(CELL_ROW)
    v = torch.from_numpy(np.ascontiguousarray(xy).view(np.uint8).reshape(len(xy), 16).copy()).to(device)
    v = parallel.gather_to_root(v, 0)
    out_xy = None if v is None else v.cpu().numpy().reshape(-1).view(np.float64).reshape(-1, 2)
    if tiles is None:
        return out_cells, out_xy
    t = torch.from_numpy(np.ascontiguousarray(tiles, dtype=np.int64).view(np.uint8).reshape(len(tiles), 8).copy()).to(device)
    return out_cells, out_xy, parallel.all_gather_records(t).cpu().numpy().reshape(-1).view(np.int64)


def _detect_artefacts(args, device):
    """GrandQC artefact contours of the slide (run by rank 0 on a background thread beside the tile loop, or inline by write_outputs)"""
    from .. import grandqc
    torch.cuda.set_device(device)
    t_a = time.time()
    logger.info("Running artefact detection")
    _, _, artefact_cnts, _ = grandqc.detect_artefacts_wsi(
        wsi.WSIReader(args.slide_path), model_art_path=args.artefact_detection_model_path,
        model_td_path=args.tissue_detection_model_path, device=device,
        tissue_override=_qc_override("tissue"), artefact_override=_qc_override("artefact"))
    logger.info(f"Found {len(artefact_cnts)} artefact contours ({time.time() - t_a:.1f} s)")
    return artefact_cnts


def _qc_override(kind: str):
    """Class-map override of a plug-in (classpose_amd/hooks.py), else None: the network's arg-max is used."""
    hooks.load_plugins()
    return hooks.qc_provider(kind) if hooks.qc_provider else None


def write_outputs(args, cells, xy, labels, plan, device=None):
    """De-duplication, ROI / tissue / artefact filters and the output files, on the cell table
    (struct of arrays): same results as the reference's list-of-dict pipeline
    (predict_wsi.py:1600-1857; ``tests/test_host_polygons_geojson.py`` checks the equivalence)."""
    logger.info(f"Number of detected cells: {len(cells)}")
    if len(cells) == 0:
        logger.warning("No cells detected")
        return None
    cen = geojson.rounded_centroids(cells)
    t_dd = time.time()
    # Identity with the reference first: the default walks scipy's own pair set in ITS order (geojson.dedup_exact).  The
    # device radius search (f2) finds the identical pair set ~10x faster but cannot know that set's iteration order, which
    # decides who survives in clusters of >= 3 cells -- so it is opt-in (CLASSPOSE_DEDUP_BACKEND=device) and says how many
    # cells it may have decided differently.
    backend = os.getenv("CLASSPOSE_DEDUP_BACKEND", "exact")
    keep = None
    if backend == "device" and device is not None:
        try:
            pairs = ops.dedup_pairs(cen, 15 / 2, device)
            keep = geojson.dedup_from_pairs(len(cells), cells["area"], pairs)
            n_od = geojson.count_order_dependent(len(cells), pairs)
            logger.info(f"De-duplication (device pair search, CLASSPOSE_DEDUP_BACKEND=device): {len(pairs)} neighbour pairs, "
                        f"{time.time() - t_dd:.2f} s")
            if n_od:
                logger.warning(f"{n_od} cells sit in clusters of >= 3 neighbours, where the reference's result depends on the "
                               "iteration order of its Python set: the fast path models that order (hash slots) and a "
                               "fraction of a percent of THESE cells can differ from the reference; unset "
                               "CLASSPOSE_DEDUP_BACKEND for the exact path")
        except Exception as e:          # e.g. the dense 8-px grid of a huge sparse slide does not fit: the exact path always works
            logger.warning(f"device de-duplication failed ({e}); falling back to the exact host path")
            keep = None
    elif backend not in ("exact", "scipy", "device"):
        raise ValueError(f"CLASSPOSE_DEDUP_BACKEND={backend!r}: expected 'exact' (default) or 'device'")
    if keep is None:
        st: dict = {}
        keep = geojson.dedup_exact(cen, cells["area"], stats=st)
        import resource
        logger.info(f"De-duplication (scipy KDTree pair set, the reference's own order): {st.get('n_pairs', 0)} neighbour pairs, "
                    f"{st.get('n_order_dependent', 0)} cells in order-dependent clusters (>= 3 neighbours), {time.time() - t_dd:.2f} s, "
                    f"peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576:.1f} GiB "
                    "(the pair set is a Python set of tuples: CLASSPOSE_DEDUP_BACKEND=device is the option for slides of many millions of cells)")
    logger.info(f"Number of cells after de-duplication: {len(keep)}")

    def filter_within(keep, polys):          # STRtree.query(points, "within"): one hit per containing polygon
        counts = np.zeros(len(keep), np.int64)
        for c in polys:
            counts += c.contains_points_strict(cen[keep, 0], cen[keep, 1])
        return np.repeat(keep, counts)

    if getattr(plan, "rois", None):
        logger.info("Filtering cells based on ROI contours")
        keep = filter_within(keep, plan.rois)
        logger.info(f"Number of cells after filtering: {len(keep)}")
    bx, by = plan.bounds
    total_tissue_area = total_artefact_area = 0
    art_shown = []                               # artefact polygons in display coordinates (after the bounds shift)
    out = Path(args.output_folder)
    out.mkdir(parents=True, exist_ok=True)
    base = Path(args.slide_path.split("?")[0]).stem if "://" not in args.slide_path else \
        args.slide_path.split("://", 1)[1].split("?")[0].replace("/", "_")
    if plan.tissue_cnts is not None:
        # predict_wsi.py:1637-1676: tissue filter + <stem>_tissue_contours.geojson (display coordinates)
        logger.info("Filtering cells based on tissue contours")
        keep = filter_within(keep, plan.tissue_cnts)
        shown = [c.translate(-bx, -by) for c in plan.tissue_cnts] if (bx != 0 or by != 0) else plan.tissue_cnts
        feats = []
        for i, cnt in enumerate(shown):
            feats.extend(roi.polygon_to_geojson(cnt, id=f"tissue_{i}", object_type="annotation",
                                                additional_properties={"classification": {"name": "tissue", "color": [0, 0, 0]}}))
        total_tissue_area = sum(c.area for c in shown)
        logger.info(f"Total tissue area: {total_tissue_area}")
        logger.info(f"Number of cells after filtering: {len(keep)}")
        with open(out / get_geojson_output_filename("tissue_contours", base), "w") as f:
            json.dump({"type": "FeatureCollection", "features": feats}, f)
    if args.artefact_detection_model_path is not None:
        if args.tissue_detection_model_path is None:
            logger.warning("Skipping artefact detection as --tissue_detection_model_path was not provided.")
        else:
            # predict_wsi.py:1678-1760: artefact detection, optional cell filter, <stem>_artefact_contours.geojson
            fut = getattr(plan, "artefact_future", None)
            t_a = time.time()
            artefact_cnts = fut.result() if fut is not None else _detect_artefacts(args, device)
            if fut is not None:
                logger.info(f"Artefact detection ran beside the tile loop; waited {time.time() - t_a:.2f} s for it here")
            art = roi.polygons_from_cnts(artefact_cnts)
            if args.filter_artefacts and art:
                hit = np.zeros(len(keep), bool)
                for a in art:
                    hit |= a.contains_points_strict(cen[keep, 0], cen[keep, 1])
                keep = keep[~hit]
                logger.info(f"Removed {int(hit.sum())} cells in artefact regions")
                logger.info(f"Cells remaining after artefact filtering: {len(keep)}")
            if bx != 0 or by != 0:
                art = [a.translate(-bx, -by) for a in art]
            feats = []
            for i, poly in enumerate(art):
                feats.extend(roi.polygon_to_geojson(poly, id=f"artefact_{i}", object_type="annotation",
                                                    additional_properties={"classification": {"name": "artefact", "color": [255, 0, 0]}}))
            art_shown = art
            total_artefact_area = sum(a.area for a in art)
            logger.info(f"Total artefact area: {total_artefact_area}")
            with open(out / get_geojson_output_filename("artefact_contours", base), "w") as f:
                json.dump({"type": "FeatureCollection", "features": feats}, f)
    if args.output_type and "csv" in args.output_type:
        # predict_wsi.py:1786-1857: per-class counts and densities over the (tissue - artefact) area
        from .. import outputs
        dens_labels = labels or ["cell"]
        names = np.array([(labels[int(c) - 1] if labels is not None else "cell") for c in cells["cls"][keep]], dtype=object)
        if plan.roi_class_dict is not None:
            prio = [c.strip() for c in args.roi_class_priority] if args.roi_class_priority else None
            # polygon centroids (unrounded; the bounds offset is applied like the reference does before this step)
            region = outputs.map_points_to_roi_classes(cells["cx"][keep] - bx, cells["cy"][keep] - by,
                                                       plan.roi_class_dict, prio)
            by_class = {k: [{"properties": {"classification": {"name": nm}}} for nm in names[region == k]]
                        for k in plan.roi_class_dict}
            df = outputs.calculate_cellular_densities(
                by_class, {k: sum(p.area for p in v) for k, v in plan.roi_class_dict.items()},
                # artefact area inside each ROI class: sum of polygon intersection areas (predict_wsi.py:1818-1828)
                {k: sum(roi.intersection_area(a, p) for a in art_shown for p in v) if args.artefact_detection_model_path else 0
                 for k, v in plan.roi_class_dict.items()},
                plan.mpp[0], plan.mpp[1], dens_labels)
        else:
            df = outputs.densities_from_counts("tissue", {l: int((names == l).sum()) for l in dens_labels},
                                               total_tissue_area, total_artefact_area, plan.mpp[0], plan.mpp[1],
                                               dens_labels)
        df.to_csv(out / f"{base}_cell_densities.csv", index=False)
        logger.info(f"Saving cellular densities to {out}/{base}_cell_densities.csv")
    contours = out / get_geojson_output_filename("cell_contours", base)
    centroids = out / get_geojson_output_filename("cell_centroids", base)
    geojson.write_feature_collections(contours, centroids, cells, xy, keep, labels, (bx, by))
    logger.info(f"Wrote {contours} and {centroids}")
    return contours, centroids


def run_main(args, spawned: bool = False):
    """One rank of ``classpose-predict-wsi`` (``predict_wsi.main`` has checked the flags and, for ``--device cuda:0,1,...``
    outside a launcher, spawned one process per listed GPU that each arrive here)."""
    if getattr(args, "inference_threads", None) is not None and (not spawned or int(os.environ.get("RANK", 0)) == 0):
        # the reference starts N Python threads per device that each call model.eval on one tile (predict_wsi.py:728-798);
        # here ONE host thread per GPU batches tiles across the whole slide into every launch, which is what those
        # threads were approximating -- the flag is accepted for drop-in compatibility and has no effect
        logger.info(f"--inference_threads {args.inference_threads} is superseded: this engine batches tiles across the slide "
                    "into each launch from one host thread per GPU (cross-tile batching replaces the per-tile eval threads)")
    devices = get_device(args.device)
    hostinfo.limit_torch_threads()
    rank, world, local = parallel.init_distributed()
    device = torch.device("cuda", local) if world > 1 else \
        (devices[0] if devices[0].index is not None else torch.device("cuda", 0))
    t_main = time.time()
    try:
        import psutil
        t_proc = psutil.Process().create_time()                   # interpreter start: the imports lie between it and here
    except Exception:                                             # noqa: BLE001
        t_proc = t_main
    if os.environ.get("CLASSPOSE_PARENT_T0"):                     # a worker of `--device cuda:0,1,...`: the command started with its parent
        t_proc = min(t_proc, float(os.environ["CLASSPOSE_PARENT_T0"]))
    cells, xy, labels, plan = run_rank(args, rank, world, device)
    t_loop_end = time.time()
    tiles = plan.cell_tiles
    if world > 1:
        n_local = len(cells)
        cells, xy, tiles = gather_cells(cells, xy, device, tiles)
        logger.info(f"[rank {rank}] exchange over {torch.distributed.get_backend()}: {n_local} local cells -> {len(cells)} cell rows on "
                    f"every rank; vertex pool {'%d vertices on rank 0' % len(xy) if xy is not None else 'sent to rank 0'}")
    if rank == 0:
        t_gather = time.time()
        cells, xy = canonical_cell_order(cells, xy, tiles)
        write_outputs(args, cells, xy, labels, plan, device)
        t_end = time.time()
        logger.info(f"wall (s): process start -> main {t_main - t_proc:.1f} (imports"
                    + (f", {os.environ['CLASSPOSE_PARENT_IMPORT_S']} s of them once in the parent the ranks were forked from" if os.environ.get("CLASSPOSE_PARENT_IMPORT_S") else "")
                    + f"), main -> end of this rank's tile loop "
                    f"{t_loop_end - t_main:.1f}, exchange {t_gather - t_loop_end:.1f}, rank-0 tail (order, de-duplication, filters, files) "
                    f"{t_end - t_gather:.1f}; total {t_end - t_proc:.1f}")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


