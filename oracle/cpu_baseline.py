"""CPU baseline worker of ``bench.py`` (test infrastructure, like the rest of ``oracle/``): the reference-shaped CPU path
on a slice of the bench workload, one tile per eval like ``predict_wsi.worker``
(/root/reference/src/classpose/entrypoints/predict_wsi.py:751-757): normalize -> run_net (4 sub-tiles in one forward,
torch-CPU fp32 ViT-L) -> compute_masks on the same injected fields -> class vote -> records.

``bench.py`` starts P of these as child processes (``python -m oracle.cpu_baseline ...``), each with its own torch
thread pool over disjoint tiles, so that all physical host cores are busy (one torch-CPU process does not scale past
~32 threads on this ViT-L: 5.1 s/tile at 32 threads, 151 s/tile at 256, measured on the GPU box's host).  Prints one
JSON line: tiles, cells, per-stage seconds and the wall-clock window [t_start, t_end] of the timed tiles.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(slide: int, first: int, stride: int, n_tiles: int, warm: int, threads: int, budget_s: float, depth: int,
        seed: int = 1234, tile: int = 256, overlap: int = 32, ncls: int = 7, start_at: float = 0.0) -> dict:
    import torch
    from classpose_amd import synth, wsi
    from oracle import classmask, dynamics, net, tiling
    torch.set_num_threads(threads)
    sd = synth.make_state_dict(ncls, None, depth=depth, seed=0)
    fw = net.make_forward(sd, torch.float32)
    coords = wsi.get_coords(tile, overlap, (slide, slide), 1.0)
    stage = {"normalise": 0.0, "network": 0.0, "dynamics": 0.0, "class_vote_records": 0.0}
    n, cells, dt = 0, 0, 0.0
    t_start = t_end = None
    for k in range(warm + n_tiles):
        (x0, y0), _ = coords[(first + k * stride) % len(coords)]
        img = synth.render_region(seed, x0, y0, tile, tile)                 # rendering is not timed
        dP, cp, lg, _ = synth.analytic_fields(seed, x0, y0, tile, tile, ncls)
        if k == warm:
            while time.time() < start_at:                                     # all workers enter the timed part together
                time.sleep(0.01)
            t_start = time.time()
        t1 = time.perf_counter()
        x = tiling.normalize_img(img[None])
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
        t_end = time.time()
        if dt > budget_s and n >= 2:
            break
    return dict(tiles=n, cells=cells, busy_s=dt, t_start=t_start, t_end=t_end, threads=threads, stage_s=stage)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--slide", type=int, default=10000)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=16)
    ap.add_argument("--warm", type=int, default=1)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--budget", type=float, default=90.0)
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--start-at", type=float, default=0.0)
    ap.add_argument("--cpus", type=str, default="", help="comma-separated logical CPU ids this worker is pinned to")
    a = ap.parse_args()
    if a.cpus:
        os.sched_setaffinity(0, {int(c) for c in a.cpus.split(",")})      # before torch creates its thread pool
    print("CPU_BASELINE " + json.dumps(run(a.slide, a.first, a.stride, a.tiles, a.warm, a.threads, a.budget, a.depth,
                                             start_at=a.start_at)), flush=True)


if __name__ == "__main__":
    main()
