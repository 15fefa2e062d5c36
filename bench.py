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
    python bench.py --slide 40000 --steps 200         # the north-star slide (31 684 tiles; ranks walk disjoint shards)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Besides the headline the line carries ``roofline`` (dominant kernel + per-stage table; the post-processing stage is timed
repetition by repetition with device AND host clocks), ``cpu_baseline`` (oracle worker processes sized by the cores / cgroup CPU quota the host grants) and, at N = 1, ``side_lines``: the same pipeline with ``--precision fp32`` and with the network's own
(random-weight) fields driving the dynamics instead of the injected ones (SURVEY 8d asks for both modes).
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

from classpose_amd import _lib, engine, hostinfo, parallel, synth, wsi  # noqa: E402
from classpose_amd.entrypoints.predict_wsi import TileStream  # noqa: E402

TILE, OVERLAP, NCLS = 256, 32, 7
N_TILES = {10000: 1936, 40000: 31684, 80000: 127449}          # SlideLoader._get_coords golden counts (tests/golden)
RESIDENT_AT_T0 = 1                                            # timed batches parked on the device when the clock starts (TileStream gate: ts.parked); the workload string and config.device_resident_batches_at_t0 both come from here
MAX_DISTINCT_BATCHES = 256                                    # rendered tiles + injected fields kept resident: 2.6 MB of fields per tile
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0              # HBM3E peak (same guide)
SEED = 1234
# algorithmic work per launch at M = 32768 tokens: GEMMs 2 M N K; attention 4 T^2 hd heads + the decomposed
# rel-pos bias 4 heads T sqrt(T) hd = 4.43 GFLOP per sub-tile (the kernel's 64-row padded tables are NOT counted)
FLOPS = {"fc1": lambda M: 2.0 * M * 4096 * 1024, "fc2": lambda M: 2.0 * M * 4096 * 1024,
         "qkv": lambda M: 2.0 * M * 3072 * 1024, "proj": lambda M: 2.0 * M * 1024 * 1024,
         "attention": lambda M: (M / 1024) * (4.0 * 1024 * 1024 * 64 * 16 + 4.0 * 16 * 1024 * 32 * 64)}
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


def physical_core_cpus() -> list[int]:
    """one logical CPU id per physical core this process may run on (sysfs topology; SMT siblings dropped)"""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as f:
                pkg = int(f.read())
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id") as f:
                core = int(f.read())
        except Exception:
            pkg, core = 0, c
        if (pkg, core) not in seen:
            seen.add((pkg, core))
            out.append(c)
    return out or allowed


def _profiler_in_env(env=None):
    """(variable, value) of the first environment entry that says this process runs UNDER a rocprofiler / roctracer tool, else None.
    Only values that name such a library count: the GPU boxes preload an unrelated guard library through LD_PRELOAD, and round 4's
    driver line lost its live traffic measurement to a test for the mere presence of that variable."""
    env = os.environ if env is None else env
    for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES"):
        v = env.get(k, "")
        if any(tag in v.lower() for tag in ("rocprof", "roctracer", "rocprofiler", "librocm-profiler", "rocprofv")):
            return k, v
    for k, v in env.items():
        if v and k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_", "ROCTRACER_")):
            return k, v
    return None


def measure_traffic_live(timeout_s: float = 90.0):
    """HBM-side bytes per launch of the dominant kernel, measured NOW on this box: two rocprofv3 --pmc passes (FETCH_SIZE,
    WRITE_SIZE -- separate passes, they do not fit one; kernel-trace only) over tools/pmc_fc1.py, which runs mlp.lin1 of one
    8-tile batch alone.  PMC counters cannot be read from inside this process, so the passes run as child processes (the
    program itself follows ``--``).  gfx950 correction of MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per 128-B request
    for wide coalesced / LDS-DMA reads -> doubled; WRITE_SIZE is exact for 16-B stores; both in KB.
    Returns (detail dict | None, reason): when the live passes do not produce a figure the reason says why (the caller falls back
    to the committed profile of the same passes and records the reason in roofline.traffic_detail)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    # never from under a profiler: a rocprofv3 started from a process that is itself running under rocprofv3 inherits the
    # outer tool's LD_PRELOAD / ROCP_TOOL_LIBRARIES, its launcher initialises the GPU and then execs -- which takes the
    # box down on this pool -- and the nested passes would pollute the outer counters anyway
    hit = _profiler_in_env()
    if hit is not None:
        return None, f"not attempted: this process runs under a profiler ({hit[0]}={hit[1][:80]})"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found on PATH or under /opt/rocm/bin"
    child_env = dict(os.environ)
    child_env["TMPDIR"] = "/tmp"
    kname = _lib.FC1_KERNEL_NAME.split(" = ")[1].split("(")[0]
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cpx_pmc_", dir="/tmp")
        try:
            # own session: on a timeout the WHOLE group goes (rocprofv3 may spawn its target rather than exec it, and an
            # orphaned pmc_fc1.py would keep the GPU busy under the side lines)
            proc = subprocess.Popen([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--", sys.executable,
                                     os.path.join(ROOT, "tools", "pmc_fc1.py")], cwd="/tmp", env=child_env,
                                    stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True, text=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
                rc = proc.returncode
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                return None, f"the rocprofv3 --pmc {counter} pass did not finish within {timeout_s:.0f} s (killed)"
            if rc != 0:
                return None, f"the rocprofv3 --pmc {counter} pass exited with code {rc}: {(err or '').strip()[-200:]}"
            tot, n = 0.0, 0
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(kname):
                            tot += float(r["Counter_Value"]); n += 1
            if n == 0:
                return None, f"the rocprofv3 --pmc {counter} pass wrote no row for {kname}"
            vals[counter] = (tot / n, n)
        except Exception as e:                                                  # noqa: BLE001 -- any failure here is a reason, not a crash
            return None, f"the rocprofv3 --pmc {counter} pass raised {type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return {"traffic": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "launches_counted": vals["FETCH_SIZE"][1]}, None


cgroup_cpu_limit = hostinfo.cgroup_cpu_limit


def cpu_baseline(slide_px, depth, n_tiles=64, warm_total=4, budget_s=75.0):
    """Reference-shaped CPU path (the oracle, kind 'port'; oracle/cpu_baseline.py) on every core the host grants the job:
    P child processes x up to 32 torch threads over disjoint tiles of the same workload (one torch-CPU process stops scaling
    at ~32 threads on this ViT-L; P and the thread count follow the physical cores of the affinity mask capped by the cgroup
    CPU quota), 64 tiles after 4 warm-up tiles unless the wall budget (75 s of timed tiles per worker, so that the default
    bench run stays within a few minutes) ends a worker earlier -- on a host that grants 16 CPUs a tile takes 5.4 s: ~14 tiles.
    The children never touch the GPU; they are started as ordinary child processes (no exec from this process)."""
    import subprocess
    cpus = physical_core_cpus()
    phys = len(cpus)
    quota = cgroup_cpu_limit()
    # what the host lets this job use: the physical cores of its affinity mask, capped by the cgroup's CPU-time quota
    # (threads beyond the quota only get throttled: 4 x 32 threads on the GPU box ran 5x slower EACH than 1 x 32)
    usable = phys if quota is None else max(1, min(phys, int(quota)))
    threads = min(32, usable)
    P = max(1, usable // threads)
    cpus = cpus[:P * threads]
    per = -(-n_tiles // P)
    warm = max(1, -(-warm_total // P))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    # each worker is pinned to its own block of physical cores (un-pinned, the four 32-thread pools of a 128-core host
    # land on each other: 25.7 s per tile instead of 4.9 s, measured)
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", "--slide", str(slide_px), "--first", str(i),
                               "--stride", str(P), "--tiles", str(per), "--warm", str(warm), "--threads", str(threads),
                               "--budget", str(budget_s), "--depth", str(depth)]
                              # several workers: each pinned to its own block of physical cores; a single worker is left to
                              # the scheduler (on a shared host the first cores are not the idle ones)
                              + (["--cpus", ",".join(map(str, cpus[i * threads:(i + 1) * threads]))] if P > 1 else []),
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(P)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=budget_s * 4 + 600)
        for ln in out.splitlines():
            if ln.startswith("CPU_BASELINE "):
                res.append(json.loads(ln[len("CPU_BASELINE "):]))
    if not res:
        return dict(value=None, unit="tiles/s", cores=phys, kind="port", sample="no worker finished")
    tiles = sum(r["tiles"] for r in res)
    cells = sum(r["cells"] for r in res)
    rate = sum(r["tiles"] / (r["t_end"] - r["t_start"]) for r in res)            # the workers' windows overlap: rates add
    span = max(r["t_end"] for r in res) - min(r["t_start"] for r in res)
    stage = {k: round(sum(r["stage_s"][k] for r in res) / tiles * 1e3, 2) for k in res[0]["stage_s"]}
    return dict(value=rate, unit="tiles/s", cores=len(res) * threads, kind="port", cells_per_s=cells / tiles * rate,
                processes=len(res), threads_per_process=threads, physical_cores=phys, cgroup_cpu_quota=quota, stage_ms_per_tile=stage,
                sample=f"{tiles} tiles of the same workload ({len(res)} process(es) x {threads} torch threads"
                       f"{', each pinned to its own block of physical cores' if len(res) > 1 else ''}; host: {phys} physical cores, cgroup CPU quota "
                       f"{quota if quota is not None else 'none'}; disjoint tiles, "
                       f"{warm} warm-up tile(s) each = {warm * len(res)} in all), one tile per eval (4 sub-tiles, fp32 torch-CPU "
                       f"ViT-L + oracle dynamics on the injected fields); all windows within {span:.1f} s; the literal "
                       f"reference cannot run here (cellpose / cv2 / openslide wheels absent)")


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N ...` started directly: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>` as a child process (this process has not initialised a GPU: torch.cuda.device_count() does not) and return its exit code.
    Fewer than N GPUs is an error unless CPX_DIST_BACKEND=gloo asks for the dry run in which several ranks share a device."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("CPX_DIST_BACKEND", "") != "gloo":
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); nothing was run "
              f"(CPX_DIST_BACKEND=gloo runs {n} ranks over the GPUs there are, as a dry run of the launch line)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def _mmm(v):
    v = sorted(v)
    return {"min": round(v[0], 4), "median": round(v[len(v) // 2], 4), "max": round(v[-1], 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = this rank's whole shard of the slide (242 steps on 1 GPU at --slide 10000)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-tiles", type=int, default=8)
    ap.add_argument("--depth", type=int, default=24)
    ap.add_argument("--slide", type=int, default=10000, help="side of the synthetic slide in pixels (10000 = configs[1]; 40000 = the north-star slide)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    ap.add_argument("--no-side-lines", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic then comes from the committed profile)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves -- as fresh child processes, BEFORE anything in this
        # process has touched a GPU -- and leave with their exit code.  (Until round 4 this ran ONE rank and printed n_gpus: 1.)
        sys.exit(spawn_ranks(args.gpus))
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would misreport n_gpus")
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))   # (dry runs may map 2 ranks to 1 GPU)
    torch.cuda.set_device(dev)
    L = _lib.lib()

    sd = synth.make_state_dict(NCLS, None, depth=args.depth, seed=0)
    w = engine.NetWeights.from_state_dict(sd, "bf16", dev)
    bt = args.batch_tiles
    eng = engine.Engine(w, TILE, batch_tiles=bt)
    S = args.slide
    slide = synth.SyntheticSlide(S, S, mpp=0.5, seed=SEED)
    plan = wsi.plan_slide(slide, TILE, OVERLAP, 0.5)
    coords = plan.coords
    assert len(coords) == N_TILES.get(S, len(coords)), (S, len(coords))
    mine = list(parallel.shard_indices(len(coords), rank, world))      # tile k -> rank k % world: disjoint shards
    shard_steps = len(mine) // bt
    steps = args.steps if args.steps > 0 else shard_steps
    # distinct batches kept resident (rendered tiles in host memory, injected fields in HBM); longer runs wrap around them
    n_distinct = min(steps + args.warmup, shard_steps, MAX_DISTINCT_BATCHES)
    use = mine[: n_distinct * bt]

    # ---- before the timed region: rendered slide tiles in host memory, analytic fields in HBM
    hostinfo.limit_torch_threads()
    with ThreadPoolExecutor(max_workers=max(2, min(32, hostinfo.usable_cpus()))) as pool:
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
    rec_keep = []

    def collect(e, sid, cells_acc, keep=False):
        out = e.result(sid)                                            # current stream waits for the post stream
        recs = out.records.view(bt, e.max_rec, rec_bytes)[:, :max_cells]
        pinned.copy_(recs, non_blocking=True)                          # records leave the device
        pinned_cnt.copy_(out.rec_counts, non_blocking=True)
        cells_acc.add_(out.nlabels.sum())
        if keep:
            rec_keep.append(recs.clone())
        # ... and the host waits for them, as the CLI's tile loop does (run_rank's collect() reads the batch's label counts back before it
        # takes the next batch): one batch runs on the device while the previous one is collected, the host is never more than one
        # step ahead.  Until round 5 this loop only QUEUED the read-backs, so the host ran ~12 steps (2 500 launches, 12 fresh tile
        # buffers) ahead at the start of a run, and the first 6 - 7 steps of the device then took 23.0 instead of 21.8 ms each
        # (BENCH_DEBUG prints the per-step device times; 8 ms of a 20-step region, nothing of a 242-step one): a start-up transient
        # of an unbounded queue that the product path does not have.
        if not os.environ.get("BENCH_UNBOUNDED_QUEUE"):
            torch.cuda.current_stream(dev).synchronize()

    def make_stream(n, first_batch, gate_at=None):
        """pinned buffers + reader threads of the CLI's TileStream over n batches, NOT started yet"""
        idxs = [use[((first_batch + i) % n_distinct) * bt + k] for i in range(n) for k in range(bt)]
        return TileStream(cached, plan, idxs, bt, TILE, TILE, dev, autostart=False, gate_at=gate_at)

    def run_steps(e, it, n, cells_acc, inject=True, keep_last=False):
        """n steps from the batch iterator `it`: TileStream (reader threads -> pinned batches -> hipMemcpyAsync on its
        copy stream) feeds the 2-stream engine pipeline; the network of step i+1 overlaps the post-processing of step i."""
        prev = None
        dbg_t = [time.perf_counter()] if os.environ.get("BENCH_DEBUG") else None
        dbg_ev = []
        if dbg_t is not None:
            dbg_ev.append(torch.cuda.Event(enable_timing=True)); dbg_ev[-1].record(torch.cuda.current_stream(dev))
        for _ in range(n):
            chunk, tiles_dev, ev, _x = next(it)
            torch.cuda.current_stream(dev).wait_event(ev)
            sid = e.submit(tiles_dev, inject=fields[batch_of[tuple(chunk)]] if inject else None, records=True)
            if dbg_t is not None:                                          # BENCH_DEBUG: when each step's network ends on the device
                dbg_ev.append(torch.cuda.Event(enable_timing=True)); dbg_ev[-1].record(e.s_net)
            if prev is not None:
                collect(e, prev, cells_acc)
            prev = sid
            if dbg_t is not None:
                dbg_t.append(time.perf_counter())
        if prev is not None:
            collect(e, prev, cells_acc, keep=keep_last)
        if dbg_t is not None:
            torch.cuda.synchronize(dev)
            dbg_t.append(time.perf_counter())
            print("BENCH_DEBUG host ms between loop iterations:", [round((b - a) * 1e3, 1) for a, b in zip(dbg_t, dbg_t[1:])], file=sys.stderr)
            print("BENCH_DEBUG device ms from the loop's start to the end of each step's network, differences:",
                  [round(a.elapsed_time(b), 2) for a, b in zip(dbg_ev, dbg_ev[1:])], file=sys.stderr)

    rec_counts: list[int] = []                # records every rank contributed to the timed run's all-gather
    decode_ahead = None                       # TileStream.ahead of the timed run: batches the reader may have decoded into pinned memory at t0

    def timed_run(e, n_steps, n_warm, inject=True, prof=None, collective=False):
        """ONE TileStream over warm-up + timed batches (its reader / copy threads pay their one-off HIP thread start-up
        during the warm-up); a gate keeps the timed batches off the device until the clock starts, except the FIRST, which is resident
        when it does (the contract's "inputs resident in HBM when the timed region starts", for one batch; every other batch is copied
        inside the region, on the copy stream, as the CLI does) -- the reader may have decoded the next few into pinned host memory by
        then, as it has at any moment of the steady state.  (Until round 4 the gate also held the decoding back: ~6 ms of GPU idle
        at the head of the region; until round 5 the first batch's copy and the reader thread's wake-up: ~3 ms.)"""
        cells_acc = torch.zeros(1, dtype=torch.int64, device=dev)
        with make_stream(n_warm + n_steps, 0, gate_at=n_warm) as ts:
            ts.start()
            it = iter(ts)
            run_steps(e, it, n_warm, cells_acc, inject)
            ts.parked.wait(timeout=60.0)                                   # the reader has issued the copy of the first timed batch
            torch.cuda.synchronize(dev)
            cells_acc.zero_()
            if prof is not None:
                e.w.c.prof = prof
            if collective:
                parallel.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ts.release()                                                   # hand-over of the first timed batch (resident on the device) and every later H2D copy happen from here on
            run_steps(e, it, n_steps, cells_acc, inject, keep_last=collective)
            allrec = None
            if collective:
                # the path's one exchange: per-cell records of this rank's shard -> every rank (RCCL)
                rec = rec_keep[-1].reshape(-1, rec_bytes)
                rec_counts[:] = parallel.all_gather_counts(rec.shape[0], dev)
                allrec = parallel.all_gather_records(rec)
            torch.cuda.synchronize(dev)
            if collective:
                parallel.barrier()
            dt = time.perf_counter() - t0
            if prof is not None:
                e.w.c.prof = None
            nonlocal decode_ahead
            decode_ahead = ts.ahead
        return dt, float(cells_acc.item()), allrec

    # the dominant GEMM is timed on every 4th layer, the sampled layers rotating by one per step so that all 24 are
    # covered equally (BENCH_PROF_STRIDE): each event pair costs the stream ~6 us of idle time on either side of the
    # launch (tools/r02_gaps.sh: 0.28 ms per step with every launch timed = 1.1 % of the headline; stride 4: +0.3-0.5 %
    # tiles/s at the same measured launch duration; stride 8 reads ~3 % longer launches on the same box -- with fewer
    # idle gaps the chip sustains a lower clock -- DESIGN 5)
    prof = C.c_void_p()
    _lib.check(L.cpx_prof_create(2 * steps * args.depth + 8, int(os.environ.get("BENCH_PROF_STRIDE", "4")), 1,
                                 C.byref(prof)), "prof_create")
    dt, cells, allrec = timed_run(eng, steps, args.warmup, inject=True, prof=prof, collective=True)
    dt = parallel.allreduce_max(dt, dev)
    cells = parallel.allreduce_sum(cells, dev)
    NK = len(_lib.PROF_KINDS)
    ms_k, cnt_k = (C.c_double * NK)(), (C.c_int * NK)()
    _lib.check(L.cpx_prof_collect(prof, ms_k, cnt_k), "prof_collect")
    L.cpx_prof_destroy(prof)
    fc1_launches = int(cnt_k[0])

    n_tiles = steps * bt * world
    M = bt * eng.n_sub * 1024
    # mlp.lin1 / mlp.lin2 run in row parts of 16 384 tokens (cpx_net_mlp_parts): one of THEIR launches covers M / parts rows
    mlp_parts = int(L.cpx_net_mlp_parts(bt * eng.n_sub, _lib.DTYPE_CODE["bf16"]))
    M_of = lambda name: M // mlp_parts if name in ("fc1", "fc2") else M
    avg_ms = ms_k[0] / max(cnt_k[0], 1)
    achieved = FLOPS["fc1"](M_of("fc1")) / (avg_ms * 1e-3) / 1e12 if fc1_launches else 0.0
    flop_per_tile = 727.3e9 * eng.n_sub
    # HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so rank 0 of
    # a 1-GPU run collects them NOW through two rocprofv3 --pmc child passes over that kernel alone (measure_traffic_live);
    # otherwise (N > 1, --no-live-traffic, rocprofv3 unavailable) the figure of the same passes over the whole bench command
    # committed under profiles/ is reported, named with its source
    traffic, traffic_src, traffic_detail = None, None, None
    fallback_reason = ("not attempted: --no-live-traffic" if args.no_live_traffic else
                       "not attempted: the live passes run on rank 0 of a 1-GPU run only" if (rank != 0 or world != 1) else None)
    if fallback_reason is None:
        traffic_detail, fallback_reason = measure_traffic_live()
        if traffic_detail is not None:
            traffic, traffic_src = traffic_detail["traffic"], "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over tools/pmc_fc1.py"
    if traffic is None:
        traffic_detail = {"fallback_reason": fallback_reason}
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    traffic = json.load(f)["traffic_bytes_per_launch"]
                traffic_src = "profiles/%s (separate rocprofv3 --pmc passes over this command)" % name
                traffic_detail["source"] = "profiles/" + name
                break
            except Exception:
                pass

    # ---- per-stage roofline (after the timed region, so the extra event pairs do not touch the headline).
    # Rounds 2 and 3 reported one bogus stage each in the driver's run (post-processing 5.03 ms, proj 0.518 ms): this pass
    # had NO warm-up -- it starts a fresh TileStream (new pinned buffers, reader threads) inside the measured steps -- and
    # reported a plain mean, so a single event pair that straddled a host stall or a clock ramp owned the figure.  Now:
    # two untimed warm-up steps on the SAME TileStream, every launch's own duration (cpx_prof_collect_launches), frac from
    # the MEDIAN, min / max beside it and an `outlier` flag when max > 3 x median.
    stages = None
    kernel_time_sum = None
    if rank == 0 and not args.no_stages:
        stages = {}
        n6 = min(6, n_distinct)
        # per layer: qkv, attention, proj + one mlp.lin1 and one mlp.lin2 launch per row part (cpx_net_mlp_parts: 2 at 8 tiles per step,
        # 3 / 4 at --batch-tiles 12 / 16); + patch embedding and neck/head spans per forward.  cpx_prof_begin drops launches past the capacity.
        stage_parts = int(L.cpx_net_mlp_parts(bt * eng.n_sub, _lib.DTYPE_CODE["bf16"]))
        cap = n6 * (args.depth * (3 + 2 * stage_parts) + 2) + 8
        prof2 = C.c_void_p()
        _lib.check(L.cpx_prof_create(cap, 1, 0x7F, C.byref(prof2)), "prof_create")
        eng.stage_timing = []
        timed_run(eng, n6, 2, inject=True, prof=prof2)
        stage_ev, eng.stage_timing = eng.stage_timing[2:], None         # (the two warm-up batches are not counted)
        ms_l, kind_l, n_l = (C.c_float * cap)(), (C.c_int * cap)(), C.c_int(0)
        _lib.check(L.cpx_prof_collect_launches(prof2, ms_l, kind_l, cap, C.byref(n_l)), "prof_collect_launches")
        L.cpx_prof_destroy(prof2)
        if n_l.value >= cap:
            raise RuntimeError("bench.py stage pass: the launch profile filled its capacity (%d): samples were dropped" % cap)
        n_l = n_l.value
        per_kind = {k: [] for k in range(len(_lib.PROF_KINDS))}
        per_fwd = args.depth * (3 + 2 * stage_parts) + 2
        where_max = {}                                        # kind -> (ms, step of the pass incl. its two warm-up steps, launch index inside the forward)
        for i in range(n_l):
            per_kind[kind_l[i]].append(float(ms_l[i]))
            if float(ms_l[i]) > where_max.get(kind_l[i], (0.0,))[0]:
                where_max[kind_l[i]] = (float(ms_l[i]), i // per_fwd, i % per_fwd)
        kernel_time_sum = 0.0
        med_of = lambda v: sorted(v)[len(v) // 2] if v else None
        # what the network stream runs besides the five per-layer kernels, and the post stream's blend: per step, median over the pass
        other = {"pre: percentile statistics + patch rows (3 launches)": med_of([e[0].elapsed_time(e[1]) for e in stage_ev]),
                 "patch_embed (1 launch)": med_of(per_kind[_lib.PROF_KINDS.index("patch_embed")]),
                 "neck + head (5 launches, one span)": med_of(per_kind[_lib.PROF_KINDS.index("neck_head")]),
                 "post stream: blend (1 launch; beside the next batch's network)": med_of([e[3].elapsed_time(e[4]) for e in stage_ev])}
        net_span = med_of([e[0].elapsed_time(e[2]) for e in stage_ev])   # first pre-processing launch -> end of the head GEMM, network stream
        for k, name in enumerate(_lib.PROF_KINDS[:5]):
            v = sorted(per_kind[k])
            if not v:
                continue
            med = v[len(v) // 2]
            tf = FLOPS[name](M_of(name)) / (med * 1e-3) / 1e12
            kernel_time_sum += med * args.depth * (M // M_of(name))
            stages[name] = {"bound": "mfma", "launch_ms": {"min": round(v[0], 4), "median": round(med, 4), "max": round(v[-1], 4)},
                            "avg_launch_ms": round(sum(v) / len(v), 4), "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "launches_timed": len(v),
                            "outlier": bool(v[-1] > 3.0 * med),
                            # where the longest launch sat: step of the stage pass (0 = the first step behind its two warm-up steps) and launch
                            # index inside that forward
                            "max_at": {"step_of_pass": where_max[k][1], "launch_in_forward": where_max[k][2], "launches_per_forward": per_fwd}}
        stages["post_processing"] = post_stage(L, eng, fields[0], bt, dev)

    # ---- side lines (N = 1): --precision fp32, and the network's own fields instead of the injected ones
    side = None
    if rank == 0 and world == 1 and not args.no_side_lines:
        side = {}
        ns = min(12, n_distinct)
        dts, cs, _ = timed_run(eng, ns, 2, inject=False)
        side["random_weight_fields_bf16"] = {
            "tiles_per_s": round(ns * bt / dts, 2), "ms_per_step": round(dts / ns * 1e3, 3), "cells_per_s": round(cs / dts, 1), "steps": ns,
            "note": "no flow injection: the dynamics consume what the random-init network itself produces (meaningless flows "
                    "-- a different number of pixels to integrate and of candidate labels than real cells give); everything else as the headline"}
        del eng
        torch.cuda.empty_cache()
        w32 = engine.NetWeights.from_state_dict(sd, "fp32", dev)
        eng32 = engine.Engine(w32, TILE, batch_tiles=bt)
        n32 = min(3, n_distinct)
        dts, cs, _ = timed_run(eng32, n32, 1, inject=True)
        side["precision_fp32"] = {
            "tiles_per_s": round(n32 * bt / dts, 2), "ms_per_step": round(dts / n32 * 1e3, 2), "cells_per_s": round(cs / dts, 1), "steps": n32,
            "network_tflops": round(n32 * bt * flop_per_tile / dts / 1e12, 1), "peak_tflops_f32_mfma": 157.3,
            "note": "--precision fp32 (what the reference's integration tests pass): exact-f32 MFMA network "
                    "(v_mfma_f32_32x32x2_f32, 1/16 of the bf16 matrix rate), same pipeline, flow injection"}
        del eng32, w32
        torch.cuda.empty_cache()
        # the reference's DEFAULT geometry (1024-px tiles, overlap 64: predict_wsi.py:92; configs[2] with puma's 10 classes) and configs[4]'s
        # 512-px fp16 tiles, host out of the way exactly as for the headline (round-5 review item 4)
        for key, a in (("configs2_geometry_1024px_puma_bf16", ("configs[2] geometry: 1024-px tiles / overlap 64, puma 10 classes", 10, 1024, 64, "bf16")),
                       ("configs4_geometry_512px_fp16", ("configs[4] geometry: 512-px tiles / overlap 32, conic 7 classes, fp16", 7, 512, 32, "fp16"))):
            try:
                side[key] = geometry_side_line(L, dev, args.depth, *a)
            except Exception as e:                                      # noqa: BLE001 -- a side line never takes the headline down
                side[key] = {"error": "%s: %s" % (type(e).__name__, e)}

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
        "build_id": _lib.build_id(),
        # what a scaling run needs to show that N ranks really exchanged over RCCL: the collective backend, the tiles every
        # rank processed in the timed steps (steps x tiles_per_step each: weak scaling) and the records each rank contributed
        # to the all-gather (their sum = config.records_gathered on every rank)
        "distributed": {"world": world, "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                        "tiles_per_rank": [steps * bt] * world, "shard_tiles_per_rank": [len(range(r, len(coords), world)) for r in range(world)],
                        "records_per_rank": list(rec_counts), "device": torch.cuda.get_device_name(dev)},
        "cells_per_sec": cells / dt,
        "network_tflops": n_tiles * flop_per_tile / dt / 1e12 / world,
        "config": {"workload": "%s: synthetic %dx%d WSI (%d tiles), tile 256 / overlap 32, "
                               "conic 7 classes, ViT-L ClassTransformer depth %d random-init, batch 32 "
                               "sub-tiles = %d WSI tiles/step, every step a distinct batch of the rank's shard "
                               "(tiles sharded k %% n_gpus, at most %d distinct batches resident, longer runs wrap) "
                               "streamed pinned host -> hipMemcpyAsync inside the "
                               "timed region (the reader decodes ahead into pinned host memory as in the steady state; the first %d timed batch(es) are resident on the device when the clock starts, every later batch is copied inside the region), flow-injection dynamics" % (
                                   "configs[1]" if S == 10000 else "north-star slide" if S == 40000 else "custom slide",
                                   S, S, len(coords), args.depth, bt, MAX_DISTINCT_BATCHES, RESIDENT_AT_T0),
                   "slide": S, "tile": TILE, "overlap": OVERLAP, "batch_subtiles": bt * 4,
                   "tiles_per_step": bt, "distinct_batches": n_distinct, "records_gathered": int(allrec.shape[0]),
                   # since round 4 the gate holds back the H2D copy of the timed batches, not their decoding: up to this many of them
                   # (of `steps`) may already sit decoded in pinned host memory when the clock starts, as at any moment of the steady
                   # state.  Rounds 1-3 also timed their decoding (worth ~1.3 % at 20 steps, nothing over the whole slide)
                   "decode_ahead_batches_at_t0": decode_ahead, "device_resident_batches_at_t0": RESIDENT_AT_T0},
        "roofline": {"bound": "mfma", "kernel": "%s (mlp.lin1 %dx4096x1024%s)" % (
                         _lib.FC1_KERNEL_NAME, M_of("fc1"), "" if mlp_parts == 1 else "; the %d token rows of a step in %d launches per layer" % (M, mlp_parts)),
                     "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                     "traffic_unit": "bytes/launch = FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; %s" % traffic_src,
                     "traffic_detail": traffic_detail,
                     "algorithmic_bytes": 2.0 * (M_of("fc1") * 1024 + 4096 * 1024 + M_of("fc1") * 4096),
                     "launches_timed": fc1_launches, "avg_launch_ms": avg_ms},
    }
    if stages is not None:
        line["roofline"]["stages"] = stages
        # the five per-layer kernels at their median launch time x depth: the part of ms_per_step the network stream spends
        # in them (must be <= ms_per_step; the rest is the small kernels: patch embedding, neck, head, blend, normalise)
        line["roofline"]["kernel_time_sum_ms_per_step"] = round(kernel_time_sum, 3)
        # ... and the rest, itemised (stage pass: every launch between its own event pair, so its steps are slower than the headline's):
        # network_stream_span = first pre-processing launch -> end of the head GEMM of one batch; span - (five kernels x depth) - the three
        # network-stream items = what no kernel owns: ~125 launch boundaries + the event pairs of this pass
        on_net = sum(v for k, v in other.items() if v is not None and not k.startswith("post stream"))
        line["roofline"]["other_kernels_ms_per_step"] = {k: (round(v, 4) if v is not None else None) for k, v in other.items()}
        line["roofline"]["other_kernels_ms_per_step"]["sum on the network stream"] = round(on_net, 4)
        if net_span is not None:
            line["roofline"]["network_stream_span_ms_per_step_stage_pass"] = round(net_span, 3)
            line["roofline"]["launch_boundaries_and_event_pairs_ms_per_step_stage_pass"] = round(net_span - kernel_time_sum - on_net, 3)
        line["roofline"]["ms_per_step_minus_itemised"] = round(dt / steps * 1e3 - kernel_time_sum - on_net, 3)
    if side is not None:
        line["side_lines"] = side
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(S, args.depth)
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def geometry_side_line(L, dev, depth, name, ncls, T, overlap, precision, n_steps=6, n_warm=2):
    """The headline's pipeline (rendered tiles in host memory -> TileStream -> 2-stream engine with flow injection -> records read back per
    step) on ANOTHER geometry of BASELINE.json: the reference's default 1024-px tile / overlap 64 (predict_wsi.py:92,1455-1458; configs[2],
    puma's 10 classes) and the 512-px fp16 tiles of configs[4].  Tiles per launch by the CLI's rule (_tile_loop.run_rank).  Reports
    sub-tiles/s (what to hold against configs[1]'s), tiles/s, cells/s and the post-processing chain alone at that tile size."""
    import math
    sd = synth.make_state_dict(ncls, None, depth=depth, seed=0)
    w = engine.NetWeights.from_state_dict(sd, precision, dev)
    del sd
    n_sub = engine.make_tiling(T, T, 256, False).ny ** 2
    nT = max(1, 96 // n_sub)
    step = 8 // math.gcd(n_sub, 8)
    nT = -(-nT // step) * step
    eng = engine.Engine(w, T, batch_tiles=nT)
    n_b = n_steps + n_warm
    per_row = math.isqrt(n_b * nT - 1) + 1
    S = (T - overlap) * per_row + overlap
    slide = synth.SyntheticSlide(S, S, mpp=0.5, seed=SEED)
    plan = wsi.plan_slide(slide, T, overlap, 0.5)
    use = list(range(n_b * nT))
    assert len(plan.coords) >= len(use) and all(plan.coords[i][1] == T for i in use)
    with ThreadPoolExecutor(max_workers=max(2, min(32, hostinfo.usable_cpus()))) as pool:
        rendered = list(pool.map(lambda ti: np.concatenate(
            [synth.render_region(SEED, plan.coords[ti][0][0], plan.coords[ti][0][1], T, T), np.full((T, T, 1), 255, np.uint8)], -1), use))
        fields_h = list(pool.map(lambda ti: synth.analytic_fields(SEED, plan.coords[ti][0][0], plan.coords[ti][0][1], T, T, ncls)[:3], use))
    cached = CachedSlide(slide, {tuple(plan.coords[ti][0]): r for ti, r in zip(use, rendered)})
    fields = [tuple(torch.from_numpy(np.stack([a[k] for a in fields_h[b * nT:(b + 1) * nT]])).to(dev) for k in range(3)) for b in range(n_b)]
    del fields_h, rendered
    rec_bytes = C.sizeof(_lib.CpxRecord)
    cells_acc = torch.zeros(1, dtype=torch.int64, device=dev)
    host_cnt = torch.empty(nT, dtype=torch.int32).pin_memory()

    def collect(sid):
        out = eng.result(sid)
        host_cnt.copy_(out.rec_counts, non_blocking=True)
        cells_acc.add_(out.nlabels.sum())
        torch.cuda.current_stream(dev).synchronize()            # as the CLI's loop: a batch's counts are read back before the next is taken
        if int(host_cnt.max()) >= eng.max_rec:
            raise RuntimeError("side line: record table overflow")

    with TileStream(cached, plan, use, nT, T, T, dev, autostart=False, gate_at=n_warm) as ts:
        ts.start()
        it = iter(ts)
        prev = None

        def run(n, b0):
            nonlocal prev
            for i in range(n):
                chunk, tiles_dev, ev, _x = next(it)
                torch.cuda.current_stream(dev).wait_event(ev)
                sid = eng.submit(tiles_dev, inject=fields[b0 + i], records=True)
                if prev is not None:
                    collect(prev)
                prev = sid
        run(n_warm, 0)
        collect(prev); prev = None
        ts.parked.wait(timeout=120.0)
        torch.cuda.synchronize(dev)
        cells_acc.zero_()
        t0 = time.perf_counter()
        ts.release()
        run(n_steps, n_warm)
        collect(prev)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    cells = float(cells_acc.item())
    post = post_chain_ms(L, eng, fields[n_warm], nT, ncls, T, dev)
    flop_sub = 727.3e9
    out = {"geometry": name, "precision": precision, "tile": T, "overlap": overlap, "classes": ncls, "subtiles_per_tile": n_sub, "tiles_per_step": nT,
           "steps": n_steps, "ms_per_step": round(dt / n_steps * 1e3, 2), "tiles_per_s": round(n_steps * nT / dt, 2),
           "subtiles_per_s": round(n_steps * nT * n_sub / dt, 1), "cells_per_s": round(cells / dt, 1), "cells_per_tile": round(cells / (n_steps * nT), 1),
           "network_tflops": round(n_steps * nT * n_sub * flop_sub / dt / 1e12, 1),
           "post_processing_ms_per_batch_alone": post["ms"], "post_processing_launches": post["launches"],
           "post_processing_share_of_step_if_serial": round(post["ms"] / (dt / n_steps * 1e3), 4),
           "note": "same pipeline as the headline (pre-rendered tiles in host memory -> TileStream -> engine, flow injection, records read back every step); "
                   "the post-processing chain runs on its own stream beside the next batch's network: ms_per_step > network alone only if it does not fit"}
    del eng, w, fields
    torch.cuda.empty_cache()
    return out


def post_chain_ms(L, eng, fields0, bt, ncls, T, dev, reps=8):
    """median device time (HIP events, back to back) of the fused post-processing chain alone on one batch of an arbitrary geometry"""
    sl = eng.slots[0]
    dP, cp, lg = fields0
    st = torch.cuda.current_stream(dev).cuda_stream

    def once():
        _lib.check(L.cpx_compute_masks_records(dP.data_ptr(), cp.data_ptr(), lg.data_ptr(), bt, ncls, T, T, 0.0, 0.4, 200, 15, 0.4,
                                               sl.masks.data_ptr(), sl.class_masks.data_ptr(), sl.nlabels.data_ptr(), eng.max_rec,
                                               sl.records.data_ptr(), sl.rec_counts.data_ptr(), sl.pp_ws.data_ptr(), st), "compute_masks_records")
    n0 = L.cpx_postproc_launch_count()
    once()
    launches = int(L.cpx_postproc_launch_count() - n0)
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); once(); e1.record()
    torch.cuda.synchronize(dev)
    v = sorted(a.elapsed_time(b) for a, b in ev)
    return {"ms": round(v[reps // 2], 4), "launches": launches}


def post_stage(L, eng, fields0, bt, dev, reps=20):
    """Post-processing alone on the GPU (blend excluded): dynamics + class vote + records of one 8-tile batch.  Every
    repetition gets its own device event pair AND host clock, in two modes -- back to back (the host runs ahead, the
    device figure is the chain's own time unless the host cannot issue fast enough) and synchronised (idle device, one
    repetition in flight: issue + drain wall) -- so that a slow host issue path and a slow device path can be told apart."""
    sl = eng.slots[0]
    dP, cp, lg = fields0
    st = torch.cuda.current_stream(dev).cuda_stream

    def once():
        _lib.check(L.cpx_compute_masks_records(dP.data_ptr(), cp.data_ptr(), lg.data_ptr(), bt, NCLS, TILE, TILE, 0.0, 0.4, 200, 15, 0.4,
                                               sl.masks.data_ptr(), sl.class_masks.data_ptr(), sl.nlabels.data_ptr(), eng.max_rec,
                                               sl.records.data_ptr(), sl.rec_counts.data_ptr(), sl.pp_ws.data_ptr(), st), "compute_masks_records")
    n0 = L.cpx_postproc_launch_count()
    once()
    launches = int(L.cpx_postproc_launch_count() - n0)
    once()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    host_issue = []
    t_all = time.perf_counter()
    for e0, e1 in ev:                                  # mode A: back to back
        e0.record()
        h0 = time.perf_counter()
        once()
        host_issue.append((time.perf_counter() - h0) * 1e3)
        e1.record()
    torch.cuda.synchronize(dev)
    wall_b2b = (time.perf_counter() - t_all) * 1e3 / reps
    dev_b2b = [a.elapsed_time(b) for a, b in ev]
    dev_sync, wall_sync = [], []
    for e0, e1 in ev:                                  # mode B: one repetition at a time on an idle device
        torch.cuda.synchronize(dev)
        h0 = time.perf_counter()
        e0.record()
        once()
        e1.record()
        torch.cuda.synchronize(dev)
        wall_sync.append((time.perf_counter() - h0) * 1e3)
        dev_sync.append(e0.elapsed_time(e1))
    ms = sorted(dev_b2b)[reps // 2]
    gbs = POST_BYTES_PER_TILE * bt / (ms * 1e-3) / 1e9
    prof_sum = None
    for name in ("r06_post_kernel_sum.json", "r05_post_kernel_sum.json", "r04_post_kernel_sum.json", "r03_post_kernel_sum.json"):    # rocprofv3 kernel-time sum of the same chain (tools/r04_profile.sh)
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                prof_sum = json.load(f)
            break
        except Exception:
            pass
    return {"bound": "hbm", "ms_per_batch": round(ms, 4), "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 5), "algorithmic_bytes_per_tile": POST_BYTES_PER_TILE,
            "launches_per_batch": launches, "reps": reps,
            "device_ms_back_to_back": _mmm(dev_b2b), "host_issue_ms_back_to_back": _mmm(host_issue),
            "wall_ms_per_rep_back_to_back": round(wall_b2b, 4),
            "device_ms_synchronised": _mmm(dev_sync), "wall_ms_synchronised": _mmm(wall_sync),
            "rocprof_kernel_time_sum": prof_sum,
            "note": "compute_masks + instance_records of one 8-tile batch alone on the GPU; ms_per_batch = median of the "
                    "back-to-back device times; a dispatch-bound chain of small kernels, hidden on the post stream in the pipeline"}


if __name__ == "__main__":
    main()
