"""CPU: host-side logic of classpose_amd (no compute calls on the HIP library)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from classpose_amd import _lib, engine, synth
from oracle import tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    so = _lib.LIB_PATH
    if not (os.path.exists(so) and os.path.exists(_lib.DEBUG_LIB_PATH)):
        _lib.build()
    L = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "classpose_hip.h")).read()
    declared = set(re.findall(r"\b(cpx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().cpx_abi_version() == _lib.ABI_VERSION
    assert _lib.lib().cpx_postproc_workspace_bytes(2, 256, 256) > 0
    # what the product library exports is EXACTLY the product header: no A/B setter, no diagnostic entry point
    import subprocess
    exported = set(re.findall(r" T (cpx_[a-z0-9_]+)", subprocess.check_output(["nm", "-D", so], text=True)))
    assert exported == declared, exported ^ declared
    # the -DCPX_DEBUG build adds include/classpose_hip_debug.h on top of the same ABI
    dbg_hdr = open(os.path.join(ROOT, "include", "classpose_hip_debug.h")).read()
    dbg_declared = set(re.findall(r"\b(cpx_[a-z0-9_]+)\s*\(", dbg_hdr))
    assert dbg_declared == set(_lib._PRIVATE), dbg_declared ^ set(_lib._PRIVATE)
    dbg_exported = set(re.findall(r" T (cpx_[a-z0-9_]+)", subprocess.check_output(["nm", "-D", _lib.DEBUG_LIB_PATH], text=True)))
    assert dbg_exported == declared | dbg_declared, dbg_exported ^ (declared | dbg_declared)
    with _lib.use_debug_library() as D:
        assert _lib.lib() is D and D.cpx_abi_version() == _lib.ABI_VERSION
    assert _lib.lib() is not D


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CpxError):
        _lib.lib()


@pytest.mark.parametrize("H,W,aug", [(256, 256, False), (256, 256, True), (512, 512, False),
                                     (300, 260, False), (1024, 1024, False), (200, 256, True)])
def test_make_tiling_matches_oracle(H, W, aug):
    t = engine.make_tiling(H, W, 256, aug)
    x = np.zeros((1, H, W, 3), np.float32)
    _, geom = tiling.subtile_batch(x, 256, aug)
    assert (t.ny, t.nx) == (geom["ny"], geom["nx"])
    assert (t.Ly, t.Lx) == (geom["Ly"], geom["Lx"])
    assert (t.ypad1, t.xpad1) == (geom["pads"][0], geom["pads"][2])
    ys = sorted({a for a, _ in geom["ysub"]})
    xs = sorted({a for a, _ in geom["xsub"]})
    assert list(t.ystart[: t.ny]) == ys and list(t.xstart[: t.nx]) == xs


def test_sub_tile_counts_of_the_baseline_configs():
    assert engine.make_tiling(256, 256).ny * engine.make_tiling(256, 256).nx == 4
    assert engine.make_tiling(256, 256, augment=True).ny ** 2 == 9
    assert engine.make_tiling(512, 512).ny ** 2 == 9
    assert engine.make_tiling(512, 512, augment=True).ny ** 2 == 25


def test_taper_matches_oracle():
    assert np.array_equal(engine.taper_1d(256), tiling.taper_mask_1d(256))
    assert np.array_equal(np.outer(engine.taper_1d(256), engine.taper_1d(256)), tiling.taper_mask(256, 256))


@pytest.mark.parametrize("n", [65536, 256 * 300, 1000, 97])
def test_percentile_params_reproduce_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.integers(0, 256, n).astype(np.uint8).astype(np.float32)
    s = np.sort(x)
    for q in (1, 99):
        prev, g = engine.percentile_params(n, q)
        a, b = s[prev], s[min(prev + 1, n - 1)]
        t = np.float32(g)
        diff = b - a
        r = a + diff * t
        if t >= 0.5:
            r = b - diff * (np.float32(1) - t)
        assert np.float32(r) == np.percentile(x, q)


def test_synthetic_slide_is_a_pure_function_of_coordinates():
    s = synth.SyntheticSlide(2000, 1500, mpp=0.5, seed=7)
    a = s.read_region((100, 200), 0, (300, 200))
    b = s.read_region((250, 300), 0, (100, 50))
    assert a.shape == (200, 300, 4) and a.dtype == np.uint8
    assert np.array_equal(a[100:150, 150:250], b)
    assert float(s.properties["openslide.mpp-x"]) == 0.5
    u = synth.SyntheticSlide.from_uri("synthetic://10000x8000?mpp=0.25&seed=3")
    assert u.level_dimensions[0] == (10000, 8000) and u.seed == 3


def test_state_dict_layout_matches_infer_structure():
    sd = synth.make_state_dict(7, None, depth=2)
    fts, ncls, depth = engine.NetWeights.infer_structure(sd)
    assert (fts, ncls, depth) == (None, 7, 2)
    sd = synth.make_state_dict(10, [64, 128], depth=1)
    fts, ncls, depth = engine.NetWeights.infer_structure(sd)
    assert (fts, ncls, depth) == ([64, 128], 10, 1)
    assert sd["W3"].shape == (640, 10, 8, 8)
    assert sd["encoder.blocks.0.attn.rel_pos_h"].shape == (27, 64)


def test_oracle_resize_linear_properties():
    """cv2.INTER_LINEAR restatement (unpinned: no cv2 here): identity, constant images,
    the 2x2 area dispatch, monotone ramps within 1 LSB of real-valued bilinear"""
    from oracle.tiling import resize_linear_u8
    rng = np.random.default_rng(3)
    t = rng.integers(0, 256, (90, 70, 3), dtype=np.uint8)
    assert np.array_equal(resize_linear_u8(t, 70, 90), t)
    c = np.full((61, 47, 3), 201, np.uint8)
    assert np.all(resize_linear_u8(c, 29, 33) == 201) and np.all(resize_linear_u8(c, 100, 120) == 201)
    a = resize_linear_u8(t, 35, 45).astype(int)
    blk = t.astype(int).reshape(45, 2, 35, 2, 3).sum((1, 3))
    assert np.array_equal(a, (blk + 2) >> 2)
    yy, xx = np.mgrid[0:100, 0:120]
    ramp = np.stack([yy * 2, xx * 2, xx + yy], -1).astype(np.uint8)
    o = resize_linear_u8(ramp, 77, 53).astype(float)
    fy = np.clip((np.arange(53) + 0.5) * 100 / 53 - 0.5, 0, 99)[:, None]
    fx = np.clip((np.arange(77) + 0.5) * 120 / 77 - 0.5, 0, 119)[None, :]
    ref = np.stack([fy * 2 + 0 * fx, fx * 2 + 0 * fy, fx + fy], -1)
    assert np.abs(o - ref).max() <= 1.0


def test_plan_slide_rescaled():
    """_init_slide arithmetic with slide mpp != model mpp (the shared manager.Value slots keep doubles)"""
    from classpose_amd import wsi
    plan = wsi.plan_slide(wsi.WSIReader("synthetic://9000x7000?mpp=0.2431"), 1024, 64, 0.5)
    scale = 0.5 / 0.2431
    rf = 1.0 / scale
    assert plan.resize_factor == rf and plan.read_tile_size == round(1024 / rf) == 2106
    assert plan.read_overlap == round(64 / rf)
    assert max(1, int(round(plan.read_tile_size * rf))) == 1024
    assert plan.coords[0] == ((0, 0), 2106) and plan.coords[1][0] == (0, 2106 - plan.read_overlap)


def test_wsi_reader_switch_imports_openslide(monkeypatch):
    """src/classpose/__init__.py:6-41: WSI_READER (default "openslide") picks the class; an OpenSlide-protocol object --
    RGBA PIL regions, string properties, several levels -- plans and reads like the reference's fill_queue
    (predict_wsi.py:220-278, 446-451); unknown reader names raise the reference's ValueError"""
    import sys
    import fake_openslide as fo
    from classpose_amd import wsi
    monkeypatch.setitem(sys.modules, "openslide", fo)
    monkeypatch.delenv("WSI_READER", raising=False)
    fo.SLIDES["/x/s.ndpi"] = dict(seed=3, base_level=2, base_dims=(600, 500), downsamples=[1.0, 4.0, 16.0],
                                  properties={"tiff.XResolution": "100000", "tiff.YResolution": "100000", "tiff.ResolutionUnit": "centimeter"})
    s = wsi.WSIReader("/x/s.ndpi")                                 # default reader = openslide
    assert isinstance(s, fo.OpenSlide) and s.level_dimensions[0] == (9600, 8000)
    plan = wsi.plan_slide(s, 512, 64, 2.0)                         # mpp 0.1 -> scale 20 -> level 2 (16x), residual 0.8
    assert plan.mpp == (0.1, 0.1) and plan.level == 2 and plan.ts == 16.0 and plan.resize_factor == 16.0 / 20.0
    assert plan.read_tile_size == 640 and plan.slide_dim == (600, 500) and plan.coords == []      # nothing fits: dropped like the reference
    plan = wsi.plan_slide(s, 256, 32, 2.0)
    assert plan.read_tile_size == 320 and [c[0] for c in plan.coords] == [(0, 0), (int(280 * 16.0), 0)]
    fo.READS.clear()
    t = wsi.read_tile(s, plan, plan.coords[1])
    assert t.shape == (320, 320, 3) and t.dtype == np.uint8 and t.flags["C_CONTIGUOUS"]
    assert fo.READS == [("/x/s.ndpi", (4480, 0), 2, (320, 320))]
    from classpose_amd import synth
    assert np.array_equal(t, synth.render_region(3, 280, 0, 320, 320))
    monkeypatch.setenv("WSI_READER", "bioformats")
    with pytest.raises(ValueError, match="not supported"):
        wsi.WSIReader("/x/s.ndpi")


@pytest.mark.parametrize("fts", [[8, 12], [20, 36], [6, 10, 12, 20]])
def test_unet_conv_list_equals_oracle_unet(fts):
    """host side of a10: ``NetWeights._build_unet_ops`` flattens unet.py:121-196 into the cpx_conv_op list, with channel
    counts padded to multiples of 8 by zero weight rows / columns.  The list, executed here on the CPU by a literal
    im2col + matmul interpreter of the op semantics (include/classpose_hip.h: cpx_conv_op), equals the oracle's UNet
    (itself pinned on the reference's UNet, tests/test_oracle_pins.py::test_unet_golden)."""
    import torch
    from oracle import net as onet
    ncls = 3
    sd = synth.make_state_dict(ncls, fts, depth=1, seed=4)
    w = engine.NetWeights.from_state_dict(sd, "fp32", "cpu")
    assert w.c.n_unet_ops == 3 * (2 * len(fts) + 2)
    by_ptr = {t.data_ptr(): t for t in w.keep}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 256, 32, 32, generator=g)
    tens = {0: x.permute(0, 2, 3, 1).contiguous()}                       # [S][h][w][c] token-major
    for i in range(w.c.n_unet_ops):
        o = w.unet_ops[i]
        W, b = by_ptr[o.weight], by_ptr[o.bias]
        a = tens[o.src_a]
        assert a.shape[-1] == o.cin_a and a.shape[1] == o.h and o.cin_a % 8 == 0 and o.cin_b % 8 == 0 and o.cout % 8 == 0
        if o.src_b >= 0:
            a = torch.cat([a, tens[o.src_b]], -1)
        S, h, _, c = a.shape
        if o.kind == 0:                                                  # conv3x3 pad 1: k = tap * c_total + c
            ap = torch.nn.functional.pad(a, (0, 0, 1, 1, 1, 1))
            col = torch.cat([ap[:, ky:ky + h, kx:kx + h] for ky in range(3) for kx in range(3)], -1)
            y = col.reshape(-1, 9 * c) @ W[:o.cout, :9 * c].T + b[:o.cout]
            y = y.reshape(S, h, h, o.cout)
        elif o.kind == 1:                                                # conv2x2 stride 2
            col = torch.cat([a[:, dy::2, dx::2] for dy in range(2) for dx in range(2)], -1)
            y = (col.reshape(-1, 4 * c) @ W[:o.cout, :4 * c].T + b[:o.cout]).reshape(S, h // 2, h // 2, o.cout)
        else:                                                            # convT2x2 stride 2: n = tap * cout + co
            z = (a.reshape(-1, c) @ W[:4 * o.cout, :c].T + b[:4 * o.cout]).reshape(S, h, h, 2, 2, o.cout)
            y = z.permute(0, 1, 3, 2, 4, 5).reshape(S, 2 * h, 2 * h, o.cout)
        tens[o.dst] = torch.relu(y) if o.relu else y
    got = tens[w.c.n_unet_ops].permute(0, 3, 1, 2)
    ref = onet.unet_forward(sd, "out_class.", x, len(fts))
    assert got.shape == ref.shape == (2, ncls * 64, 32, 32)
    assert float((got - ref).abs().max()) < 1e-4 * float(ref.abs().max())


def test_canonical_cell_order_makes_rank_count_invisible():
    """predict_wsi.canonical_cell_order: whatever order ranks / batches / the fallback path deliver cells in, the table that
    reaches de-duplication is in tile order (label order inside a tile), vertices following their cells"""
    from classpose_amd.entrypoints.predict_wsi import CELL_ROW, canonical_cell_order
    rng = np.random.default_rng(0)
    n_tiles, per = 7, [3, 0, 5, 2, 4, 1, 6]
    tiles = np.repeat(np.arange(n_tiles), per)
    n = len(tiles)
    cells = np.zeros(n, CELL_ROW)
    cells["n_pts"] = rng.integers(3, 9, n)
    cells["area"] = np.arange(n)                                   # identifies the cell
    xy = np.repeat(np.arange(n, dtype=np.float64), cells["n_pts"])[:, None] * np.ones((1, 2))     # vertex rows carry their cell's id
    c1, x1 = canonical_cell_order(cells, xy, tiles)
    assert c1 is cells and x1 is xy                                 # already canonical: untouched
    # two-rank delivery: rank 0 holds tiles 0, 2, 4, 6, rank 1 tiles 1, 3, 5
    order = np.concatenate([np.flatnonzero(tiles % 2 == 0), np.flatnonzero(tiles % 2 == 1)])
    offs = np.concatenate([[0], np.cumsum(cells["n_pts"])])
    xy2 = np.concatenate([xy[offs[i]:offs[i + 1]] for i in order])
    c2, x2 = canonical_cell_order(cells[order], xy2, tiles[order])
    assert np.array_equal(c2, cells) and np.array_equal(x2, xy)
    c3, x3 = canonical_cell_order(cells[:0], xy[:0], tiles[:0])
    assert len(c3) == 0 and len(x3) == 0


def test_library_reports_the_sources_it_was_built_from():
    """cpx_build_id() = hash of csrc/* + the two headers at build time; _lib.source_build_id() recomputes it from the files on
    disk -- a stale prebuilt .so (they are git-ignored and travel with snapshots) fails here instead of silently running."""
    from classpose_amd import _lib
    L = _lib.lib()
    assert L.cpx_build_id().decode() == _lib.source_build_id()


def test_dominant_kernel_name_the_bench_searches_for_is_a_kernel_of_the_product_library():
    """bench.py's live roofline.traffic passes (rocprofv3 --pmc over tools/pmc_fc1.py) and tools/r05_make_profiles.py pick the
    dominant kernel's rows by the name rocprofv3 prints, _lib.FC1_KERNEL_NAME.  When the fp16 instantiation gave the kernel a third
    template argument the name went stale and the driver line fell back to the committed profile (with its reason, but fell back):
    the name must be a kernel symbol of the library as built."""
    import subprocess
    name = _lib.FC1_KERNEL_NAME.split(" = ")[1]
    so = os.path.join(os.path.dirname(_lib.__file__), "libclasspose_hip.so")
    syms = subprocess.run(["nm", "-C", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert re.search(r"\b[VW] " + re.escape(name) + r"$", syms, re.M), name
    src = open(os.path.join(os.path.dirname(os.path.dirname(_lib.__file__)), "tools", "r05_make_profiles.py")).read()
    assert name.split("(")[0] in src


def test_isa_lint_no_lds_load_is_consumed_before_its_wait():
    """tools/lint_isa.py on the built libraries (CPU only: llvm-objdump of the bundled gfx950 code objects): no instruction reads the
    destination of an LDS load before an s_waitcnt lgkmcnt that covers it.  The inline-asm fragment reads of the hot kernels are only kept
    behind their inline-asm waits by statement order / sched_barrier / "+v" ties; round 4 found one that was not (the peeled last key tile
    of the attention kernel: outputs wrong by 6 % on average)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lint_isa", os.path.join(root, "tools", "lint_isa.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    if not os.path.exists(lint.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    for name in ("libclasspose_hip.so", "libclasspose_hip_debug.so"):
        path = os.path.join(root, "classpose_amd", name)
        assert os.path.exists(path), path
        found = lint.findings(path)
        assert not found, found[:5]


def test_hostinfo_caps_thread_pools_by_the_cgroup_quota(tmp_path, monkeypatch):
    """hostinfo.usable_cpus = affinity mask capped by the cgroup CPU quota (a 256-CPU box with cpu.max = 1600000 100000 grants 16 cores)."""
    import builtins
    from classpose_amd import hostinfo
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text(fake_open.content)
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    fake_open.content = "1600000 100000\n"
    assert hostinfo.cgroup_cpu_limit() == 16.0 and hostinfo.usable_cpus() == 16
    fake_open.content = "max 100000\n"
    assert hostinfo.cgroup_cpu_limit() is None and hostinfo.usable_cpus() == 256
    fake_open.content = "50000 100000\n"                      # half a core still gets one thread
    assert hostinfo.usable_cpus() == 1


def test_bench_profiler_guard_looks_at_values_not_names():
    """bench.py's live-traffic pass must not be switched off by an unrelated LD_PRELOAD (round 4's driver line lost its measurement to the
    guard library the GPU boxes preload); it is switched off under rocprofv3, and says which variable told it so."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench._profiler_in_env({"LD_PRELOAD": "/usr/lib/libexecguard.so", "PATH": "/usr/bin"}) is None
    assert bench._profiler_in_env({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})[0] == "LD_PRELOAD"
    assert bench._profiler_in_env({"ROCPROF_OUTPUT_PATH": "/tmp/x"})[0] == "ROCPROF_OUTPUT_PATH"
    assert bench._profiler_in_env({"HSA_TOOLS_LIB": "libroctracer64.so"})[0] == "HSA_TOOLS_LIB"
    assert bench._profiler_in_env({"ROCP_TOOL_LIBRARIES": ""}) is None


def test_bench_gpus_n_without_gpus_exits_nonzero():
    """`python bench.py --gpus 2` on a node without two GPUs (and no gloo dry run asked for) must not print a line at all: exit code 2."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CPX_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-300:])
    assert "--gpus 2" in r.stderr and r.stdout.strip() == ""


def test_classpose_model_engine_pool_is_bounded(monkeypatch):
    """ClassposeModel hands engines out of a bounded pool per (shape, options): at most max_engines exist however many threads call, a lease
    returns its engine on exit (also when the body raises), and a failed construction gives its reservation back."""
    import threading
    import time
    from classpose_amd import models
    made = []

    class FakeEngine:
        def __init__(self, *a, **k):
            made.append(self)
    m = models.ClassposeModel.__new__(models.ClassposeModel)
    m.max_engines, m._pool, m._cv, m.weights, m.max_batch_tiles = 2, {}, threading.Condition(), None, 8
    monkeypatch.setattr(models.engine, "Engine", FakeEngine)
    peak, live, lock = [0], [0], threading.Lock()

    def work():
        for _ in range(5):
            with m._engine(256, 256, False, 0.1, {"niter": 200}):
                with lock:
                    live[0] += 1
                    peak[0] = max(peak[0], live[0])
                time.sleep(0.002)
                with lock:
                    live[0] -= 1
    th = [threading.Thread(target=work) for _ in range(8)]
    for t in th: t.start()
    for t in th: t.join()
    assert len(made) == 2 and peak[0] == 2 and m.engines_alive() == 2
    with pytest.raises(RuntimeError):
        with m._engine(256, 256, False, 0.1, {"niter": 200}):
            raise RuntimeError("body failed")
    assert len(m._pool[next(iter(m._pool))]["idle"]) == 2            # returned
    monkeypatch.setattr(models.engine, "Engine", lambda *a, **k: (_ for _ in ()).throw(MemoryError("no memory")))
    with pytest.raises(MemoryError):
        with m._engine(512, 512, False, 0.1, {"niter": 200}):
            pass
    assert m.engines_alive() == 2                                     # the failed reservation was given back


def test_predict_wsi_front_is_free_of_torch_and_forwards_the_tile_loop_names():
    """Round 6: ``entrypoints/predict_wsi.py`` is the light front (flags, flag checks, the parent that spawns one worker per listed GPU,
    predict_wsi.py:1542-1572 of the reference); importing it and building the parser must not import torch -- the parent of a multi-GPU
    run used to pay 1 - 1.6 s for it before its children paid it again -- while every tile-loop name still resolves through it."""
    import subprocess
    code = ("import sys; from classpose_amd.entrypoints import predict_wsi as p; a = p.build_parser().parse_args(['--model_config', 'conic', "
            "'--slide_path', 's', '--output_folder', 'o', '--device', 'cuda:0,1,2']); assert p._device_ids(a.device) == [0, 1, 2]; "
            "assert p._device_ids('cuda') is None and p._device_ids(None) is None; assert 'torch' not in sys.modules, 'torch imported by the front'; "
            "assert p.DEFAULT_TILE_SIZE == 1024 and p.get_geojson_output_filename('roi', 'x') == 'x_roi.geojson'; "
            "t = p.TileStream; assert 'torch' in sys.modules and t.__module__.endswith('_tile_loop') and callable(p.run_rank) and p.CELL_ROW.itemsize == 48")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    with pytest.raises(AttributeError):
        from classpose_amd.entrypoints import predict_wsi
        predict_wsi.no_such_name


def test_spawned_workers_failure_stops_the_other_ranks_and_is_reported():
    """``--device cuda:0,1`` outside a launcher: the parent spawns one process per listed GPU (plain multiprocessing, spawn context).  A rank that fails
    -- here every rank, on the flag check, before anything touches a GPU -- ends the run with a RuntimeError naming the rank and its exit code."""
    import argparse
    from classpose_amd.entrypoints import predict_wsi
    args = argparse.Namespace(model_config="conic", slide_path="s", output_folder="o", tissue_detection_model_path=None, output_type=None,
                              tile_size=64, device="cuda:0,1")
    good = argparse.Namespace(**{**vars(args), "tile_size": 256})
    predict_wsi._check_unsupported(good)
    with pytest.raises(ValueError, match="Tile size must be at least 256"):
        predict_wsi.main(args)                                   # the parent checks the flags before it spawns anything
    with pytest.raises(RuntimeError, match=r"classpose-rank\d \(pid \d+\) exited with code 1"):
        predict_wsi._spawn_workers(args, [0, 1])


def test_half_precision_weights_need_the_device():
    """NetWeights rounds and folds the checkpoint with HIP kernels (csrc/cpx_weights.hip): a CPU device is accepted for float32 only (the host-side
    packing test above); half precision fails loudly instead of converting on the host."""
    from classpose_amd import _lib
    sd = synth.make_state_dict(3, None, depth=1, seed=1)
    with pytest.raises(_lib.CpxError, match="no CPU path"):
        engine.NetWeights.from_state_dict(sd, "bf16", "cpu")


def test_worker_start_method_is_fork_only_from_a_clean_parent(monkeypatch):
    """``--device cuda:0,1,...``: the per-GPU workers are forked from the parent (which then imports the tile loop ONCE for all of them) only while the
    parent has a single thread and no GPU context; ``CLASSPOSE_START_METHOD`` overrides; a fresh interpreter chooses fork."""
    import subprocess
    from classpose_amd.entrypoints import predict_wsi
    monkeypatch.setenv("CLASSPOSE_START_METHOD", "spawn")
    assert predict_wsi._start_method() == "spawn"
    monkeypatch.setenv("CLASSPOSE_START_METHOD", "fork")
    assert predict_wsi._start_method() == "fork"
    monkeypatch.delenv("CLASSPOSE_START_METHOD")
    import threading
    ev = threading.Event()
    t = threading.Thread(target=ev.wait)
    t.start()
    try:
        assert predict_wsi._start_method() == "spawn"            # another thread is alive: its locks would not survive a fork
    finally:
        ev.set(); t.join()
    r = subprocess.run([sys.executable, "-c", "from classpose_amd.entrypoints import predict_wsi as p; print(p._start_method())"],
                       cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "fork", (r.stdout, r.stderr[-500:])
