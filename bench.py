#!/usr/bin/env python3
"""Hot-path benchmark: WSI tiles/s (+ cells/s) of the classpose tile path on MI355X.

Workload = BASELINE.json configs[1]: synthetic 10 000 x 10 000 WSI, tile 256 / overlap 32,
conic (7 classes), bf16, batch 32 sub-tiles (= 8 WSI tiles of 4 overlapping 256^2 sub-tiles,
exactly the reference's per-tile sub-tiling, batched across tiles).  A "step" is one pass of
the whole path over one batch of 8 DISTINCT tiles of this rank's shard of the slide (tile
k -> rank k % n_gpus, x-major ``_get_coords`` order): pinned host batch -> hipMemcpyAsync on the
copy stream of the CLI's own ``TileStream`` (INSIDE the timed region) -> percentile
normalisation -> pad/sub-tile -> ViT-L ClassTransformer (24 blocks, random-init weights of the
reference layout) -> pixel-shuffle/taper blend -> flow dynamics -> instance ids -> class vote ->
per-cell records (+ their D2H copy).  With no flags the timed region is the whole slide
(242 steps = 1936 tiles).  What happens BEFORE the timed region: the procedural slide is
rendered into host memory (the stand-in for OpenSlide's JPEG decode, which is the reference's
own CPU library and out of scope) and -- because random weights produce no meaningful cells --
the analytic flow / cellprob / logit fields of the same procedural nuclei are placed in HBM:
the dynamics consume those ("flow injection") while the network still runs on the
streamed pixels.

    python bench.py                                   # 1 GPU, the whole 10k x 10k slide
    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from classpose_amd import _lib, engine, parallel, synth, wsi  # noqa: E402
from classpose_amd.entrypoints.predict_wsi import TileStream  # noqa: E402

SLIDE = 10000
TILE, OVERLAP, NCLS = 256, 32, 7
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0              # HBM3E peak (same guide)
SEED = 1234
# algorithmic work per launch at M = 32768 tokens: GEMMs 2 M N K, attention 4 T^2 hd heads + rel-pos
FLOPS = {"fc1": lambda M: 2.0 * M * 4096 * 1024, "fc2": lambda M: 2.0 * M * 4096 * 1024,
         "qkv": lambda M: 2.0 * M * 3072 * 1024, "proj": lambda M: 2.0 * M * 1024 * 1024,
         "attention": lambda M: (M / 1024) * (4.0 * 1024 * 1024 * 64 * 16 + 4.0 * 16 * 1024 * 64 * 64)}
POST_BYTES_PER_TILE = 524288 + 262144 + NCLS * 262144 + 131072 + 65536      # dP + cellprob + logits in, ids + classes out


class CachedSlide:
    """OpenSlide-protocol view of the procedural slide whose tiles were rendered ahead of the timed region
    (what a decoded-tile cache in host memory looks like to ``TileStream``)."""

    def __init__(self, slide, tiles: dict):
        self.properties, self.level_dimensions = slide.properties, slide.level_dimensions
        self.level_downsamples, self.level_count, self.seed = slide.level_downsamples, 1, slide.seed
        self._tiles = tiles

    def get_best_level_for_downsample(self, d):
        return 0

    def read_region(self, location, level, size):
        return self._tiles[(int(location[0]), int(location[1]))]


def cpu_baseline(sd, coords, n_tiles=16, warm=2, budget_s=75.0):
    """Reference-shaped CPU path (the oracle, kind 'port'), one tile per eval like
    predict_wsi.worker: normalize -> run_net (4 sub-tiles in one forward, torch-CPU fp32) ->
    compute_masks on the same injected fields -> class vote -> records."""
    from oracle import classmask, dynamics, net, tiling
    # torch-CPU scales to ~16-32 threads on this ViT-L and collapses beyond (measured on the
    # 256-core GPU-box host: 5.1 s/tile at 32 threads, 151 s/tile at 256)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    fw = net.make_forward(sd, torch.float32)
    n, cells, dt = 0, 0, 0.0
    stage = {"normalise": 0.0, "network": 0.0, "dynamics": 0.0, "class_vote_records": 0.0}
    for k in range(warm + n_tiles):
        (x0, y0), _ = coords[k]
        tile = synth.render_region(SEED, x0, y0, TILE, TILE)           # rendering is not timed
        dP, cp, lg, _ = synth.analytic_fields(SEED, x0, y0, TILE, TILE, NCLS)
        t1 = time.perf_counter()
        x = tiling.normalize_img(tile[None])
        t2 = time.perf_counter()
        tiling.run_net(fw, x, batch_size=8, bsize=256)
        t3 = time.perf_counter()
        m = dynamics.compute_masks(dP, cp)
        t4 = time.perf_counter()
        cm, _ = classmask.compute_class_masks(m, lg)
        classmask.instance_records(m, cm)
        t5 = time.perf_counter()
        if k < warm:
            continue
        dt += t5 - t1
        for key, v in zip(stage, (t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            stage[key] += v
        cells += int(m.max())
        n += 1
        if dt > budget_s and n >= 8:
            break
    return dict(value=n / dt, unit="tiles/s", cores=cores, kind="port",
                cells_per_s=cells / dt, stage_ms_per_tile={k: round(v / n * 1e3, 2) for k, v in stage.items()},
                sample=f"{n} tiles of the same workload after {warm} warm-up tiles, one tile per eval (4 sub-tiles, "
                       f"fp32 torch-CPU ViT-L on {cores} threads + oracle dynamics on the injected fields), {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = this rank's whole shard of the slide (242 steps on 1 GPU)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-tiles", type=int, default=8)
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    args = ap.parse_args()

    rank, world, local = parallel.init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))   # (dry runs may map 2 ranks to 1 GPU)
    torch.cuda.set_device(dev)
    L = _lib.lib()

    sd = synth.make_state_dict(NCLS, None, depth=args.depth, seed=0)
    w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
    bt = args.batch_tiles
    eng = engine.Engine(w, TILE, batch_tiles=bt)
    slide = synth.SyntheticSlide(SLIDE, SLIDE, mpp=0.5, seed=SEED)
    plan = wsi.plan_slide(slide, TILE, OVERLAP, 0.5)
    coords = plan.coords
    assert len(coords) == 1936
    mine = list(parallel.shard_indices(len(coords), rank, world))      # tile k -> rank k % world
    shard_steps = len(mine) // bt
    steps = args.steps if args.steps > 0 else shard_steps
    n_distinct = min(steps + args.warmup, shard_steps)                 # distinct batches; longer runs wrap around the shard
    use = mine[: n_distinct * bt]

    # ---- before the timed region: rendered slide tiles in host memory, analytic fields in HBM
    with ThreadPoolExecutor(max_workers=max(2, min(32, os.cpu_count() or 4))) as pool:
        rendered = list(pool.map(lambda ti: np.concatenate(
            [synth.render_region(SEED, coords[ti][0][0], coords[ti][0][1], TILE, TILE),
             np.full((TILE, TILE, 1), 255, np.uint8)], -1), use))
        fields_h = list(pool.map(lambda ti: synth.analytic_fields(SEED, coords[ti][0][0], coords[ti][0][1], TILE, TILE, NCLS)[:3], use))
    cached = CachedSlide(slide, {tuple(coords[ti][0]): r for ti, r in zip(use, rendered)})
    fields = []
    for b in range(n_distinct):
        f = fields_h[b * bt:(b + 1) * bt]
        fields.append(tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(dev) for k in range(3)))
    del fields_h, rendered
    batch_of = {tuple(use[b * bt:(b + 1) * bt]): b for b in range(n_distinct)}

    rec_bytes = C.sizeof(_lib.CpxRecord)
    max_cells = 256                                                    # >> the ~81 cells of a 256 px tile of this slide
    pinned = torch.empty((bt, max_cells, rec_bytes), dtype=torch.uint8).pin_memory()
    pinned_cnt = torch.empty(bt, dtype=torch.int32).pin_memory()
    cells_acc = torch.zeros(1, dtype=torch.int64, device=dev)
    rec_keep = []

    def collect(sid, keep=False):
        out = eng.result(sid)                                          # current stream waits for the post stream
        recs = out.records.view(bt, eng.max_rec, rec_bytes)[:, :max_cells]
        pinned.copy_(recs, non_blocking=True)                          # records leave the device
        pinned_cnt.copy_(out.rec_counts, non_blocking=True)
        cells_acc.add_(out.nlabels.sum())
        if keep:
            rec_keep.append(recs.clone())

    def make_stream(n, first_batch, gate_at=None):
        """pinned buffers + reader threads of the CLI's TileStream over n batches, NOT started yet"""
        idxs = [use[((first_batch + i) % n_distinct) * bt + k] for i in range(n) for k in range(bt)]
        return TileStream(cached, plan, idxs, bt, TILE, TILE, dev, autostart=False, gate_at=gate_at)

    def run_steps(it, n, keep_last=False):
        """n steps from the batch iterator `it`: TileStream (reader threads -> pinned batches -> hipMemcpyAsync on its
        copy stream) feeds the 2-stream engine pipeline; the network of step i+1 overlaps the post-processing of step i."""
        prev = None
        dbg_t = [time.perf_counter()] if os.environ.get("BENCH_DEBUG") else None
        for _ in range(n):
            chunk, tiles_dev, ev, _x = next(it)
            torch.cuda.current_stream(dev).wait_event(ev)
            sid = eng.submit(tiles_dev, inject=fields[batch_of[tuple(chunk)]], records=True)
            if prev is not None:
                collect(prev)
            prev = sid
            if dbg_t is not None:
                dbg_t.append(time.perf_counter())
        if prev is not None:
            collect(prev, keep=keep_last)
        if dbg_t is not None:
            torch.cuda.synchronize(dev)
            dbg_t.append(time.perf_counter())
            print("BENCH_DEBUG host ms between loop iterations:", [round((b - a) * 1e3, 1) for a, b in zip(dbg_t, dbg_t[1:])], file=sys.stderr)

    # ONE TileStream over warm-up + timed batches (its reader / copy threads pay their one-off HIP thread start-up
    # during the warm-up); a gate keeps it from reading or copying any timed batch before the clock starts
    ts = make_stream(args.warmup + steps, 0, gate_at=args.warmup)
    ts.start()
    it = iter(ts)
    run_steps(it, args.warmup)
    torch.cuda.synchronize(dev)
    cells_acc.zero_()
    # the dominant GEMM is timed on every 4th layer, the sampled layers rotating by one per step so that all 24 are
    # covered equally (BENCH_PROF_STRIDE): each event pair costs the stream ~6 us of idle time on either side of the
    # launch (tools/r02_gaps.sh: 0.28 ms per step with every launch timed = 1.1 % of the headline; stride 4: +0.3-0.5 %
    # tiles/s at the same measured launch duration; stride 8 reads ~3 % longer launches on the same box -- with fewer
    # idle gaps the chip sustains a lower clock -- DESIGN 5)
    prof = C.c_void_p()
    _lib.check(L.cpx_prof_create(steps * args.depth + 8, int(os.environ.get("BENCH_PROF_STRIDE", "4")), 1,
                                 C.byref(prof)), "prof_create")
    w.c.prof = prof
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ts.release()                                                       # first read / H2D copy of a timed batch happens from here on
    run_steps(it, steps, keep_last=True)
    # the path's one exchange: per-cell records of this rank's shard -> every rank (RCCL)
    rec = rec_keep[-1].reshape(-1, rec_bytes)
    allrec = parallel.all_gather_records(rec)
    torch.cuda.synchronize(dev)
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.allreduce_max(dt, dev)
    cells = parallel.allreduce_sum(float(cells_acc.item()), dev)
    ms_k, cnt_k = (C.c_double * 5)(), (C.c_int * 5)()
    _lib.check(L.cpx_prof_collect(prof, ms_k, cnt_k), "prof_collect")
    w.c.prof = None
    L.cpx_prof_destroy(prof)
    fc1_launches = int(cnt_k[0])

    n_tiles = steps * bt * world
    M = bt * eng.n_sub * 1024
    avg_ms = ms_k[0] / max(cnt_k[0], 1)
    achieved = FLOPS["fc1"](M) / (avg_ms * 1e-3) / 1e12 if fc1_launches else 0.0
    flop_per_tile = 727.3e9 * eng.n_sub
    traffic, traffic_src = None, None
    for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:      # HBM-side bytes per launch of the dominant kernel: separate rocprofv3 --pmc passes, committed
            with open(os.path.join(ROOT, "profiles", name)) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"]
            traffic_src = name
            break
        except Exception:
            pass

    # ---- per-stage roofline (after the timed region, so the extra event pairs do not touch the headline)
    stages = None
    if rank == 0 and not args.no_stages:
        stages = {}
        prof2 = C.c_void_p()
        _lib.check(L.cpx_prof_create(6 * args.depth * 5 + 8, 1, 0x1F, C.byref(prof2)), "prof_create")
        w.c.prof = prof2
        n6 = min(6, n_distinct)
        ts2 = make_stream(n6, 0)
        ts2.start()
        run_steps(iter(ts2), n6)
        torch.cuda.synchronize(dev)
        _lib.check(L.cpx_prof_collect(prof2, ms_k, cnt_k), "prof_collect")
        w.c.prof = None
        L.cpx_prof_destroy(prof2)
        for k, name in enumerate(_lib.PROF_KINDS):
            if cnt_k[k]:
                ms = ms_k[k] / cnt_k[k]
                tf = FLOPS[name](M) / (ms * 1e-3) / 1e12
                stages[name] = {"bound": "mfma", "avg_launch_ms": round(ms, 4), "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
                                "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "launches_timed": cnt_k[k]}
        # post-processing (blend excluded): dynamics + class vote + records of one 8-tile batch, alone on the GPU
        sl = eng.slots[0]
        dP, cp, lg = fields[0]
        st = torch.cuda.current_stream(dev).cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        for r in range(reps + 2):
            if r == 2:
                e0.record()
            _lib.check(L.cpx_compute_masks(dP.data_ptr(), cp.data_ptr(), lg.data_ptr(), bt, NCLS, TILE, TILE, 0.0, 0.4, 200, 15, 0.4,
                                           sl.masks.data_ptr(), sl.class_masks.data_ptr(), sl.nlabels.data_ptr(),
                                           sl.pp_ws.data_ptr(), st), "compute_masks")
            _lib.check(L.cpx_instance_records(sl.masks.data_ptr(), sl.class_masks.data_ptr(), bt, TILE, TILE, eng.max_rec,
                                              sl.records.data_ptr(), sl.rec_counts.data_ptr(), sl.pp_ws.data_ptr(), st), "records")
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        gbs = POST_BYTES_PER_TILE * bt / (ms * 1e-3) / 1e9
        stages["post_processing"] = {"bound": "hbm", "ms_per_batch": round(ms, 4), "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS,
                                     "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 5),
                                     "algorithmic_bytes_per_tile": POST_BYTES_PER_TILE,
                                     "note": "compute_masks + instance_records of one 8-tile batch alone on the GPU; latency-bound "
                                             "chain of small kernels, hidden on the post stream in the pipeline"}

    line = {
        "metric": "wsi_tiles_per_sec",
        "value": n_tiles / dt,
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "cells_per_sec": cells / dt,
        "network_tflops": n_tiles * flop_per_tile / dt / 1e12 / world,
        "config": {"workload": "configs[1]: synthetic 10000x10000 WSI (1936 tiles), tile 256 / overlap 32, "
                               "conic 7 classes, ViT-L ClassTransformer depth %d random-init, batch 32 "
                               "sub-tiles = %d WSI tiles/step, every step a distinct batch of the rank's shard "
                               "(tiles sharded k %% n_gpus) streamed pinned host -> hipMemcpyAsync inside the "
                               "timed region, flow-injection dynamics" % (args.depth, bt),
                   "tile": TILE, "overlap": OVERLAP, "batch_subtiles": bt * eng.n_sub,
                   "tiles_per_step": bt, "distinct_batches": n_distinct, "records_gathered": int(allrec.shape[0])},
        "roofline": {"bound": "mfma", "kernel": "k_gemm256p<GELU> = void k_gemm256p<1, false, 1>(GemmArgs) (mlp.lin1 %dx4096x1024)" % M,
                     "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                     "traffic_unit": "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % traffic_src,
                     "algorithmic_bytes": 2.0 * (M * 1024 + 4096 * 1024 + M * 4096),
                     "launches_timed": fc1_launches, "avg_launch_ms": avg_ms},
    }
    if stages is not None:
        line["roofline"]["stages"] = stages
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, coords)
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
