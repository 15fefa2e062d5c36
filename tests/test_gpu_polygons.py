"""GPU: device polygonisation (cpx_polygonize_device, f1) against the CPU oracle (oracle/polygons.py: Suzuki-Abe
border following in OpenCV's contour order + GEOS ring metrics, predict_wsi.py:601-652) and, bit for bit, against
the product's host polygoniser (cpx_polygonize_host, itself oracle-checked in tests/test_oracle_polygons.py)."""
import numpy as np
import pytest
import torch

from classpose_amd import engine, postprocess, synth
from oracle import polygons as opoly

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,nT,scale", [(256, 4, 1.0), (320, 2, 2.2727), (1024, 1, 1.0)])
def test_device_polygons_equal_host(cuda, T, nT, scale):
    sd = synth.make_state_dict(7, None, depth=1, seed=1)
    w = engine.NetWeights.from_state_dict(sd, "bf16", cuda)
    eng = engine.Engine(w, T, batch_tiles=nT)
    origins = [(1000 * i + 17, 333 * i + 5) for i in range(nT)]
    tiles = np.stack([synth.render_region(9, x, y, T, T) for x, y in origins])
    f = [synth.analytic_fields(9, x, y, T, T, 7) for x, y in origins]
    inj = tuple(torch.from_numpy(np.stack([a[k] for a in f])).to(cuda) for k in range(3))
    out = eng.run(torch.from_numpy(tiles).to(cuda), inject=inj, records=True, polygons=(scale, origins))
    got = eng.fetch_polygons(nT, out)
    assert got is not None
    cells, tile, xy = got
    masks = out.masks.cpu().numpy().view(np.uint16)
    recs = eng.fetch_records(nT, out)
    n_seen = 0
    for t in range(nT):
        hc, hxy = postprocess.polygonize_tile(masks[t], recs[recs["tile"] == t], scale, origins[t])
        dc = cells[tile == t]
        assert len(dc) == len(hc) > (20 if T == 256 else 50)
        for name in ("area", "perimeter", "cx", "cy", "n_pts", "valid", "cls"):
            assert np.array_equal(dc[name], hc[name]), (t, name)
        for a, b in zip(dc, hc):
            assert np.array_equal(xy[a["offset"]: a["offset"] + a["n_pts"]], hxy[b["offset"]: b["offset"] + b["n_pts"]])
        n_seen += len(dc)
        # the independent oracle: PostProcessor.__call__ restated (valid cells only, label order)
        cm = out.class_masks[t].cpu().numpy()
        ref = opoly.post_process_tile(masks[t], cm, origins[t], scale)
        dv = dc[dc["valid"] == 1]
        assert len(dv) == len(ref)
        for a, r in zip(dv, ref):
            assert np.array_equal(xy[a["offset"]: a["offset"] + a["n_pts"]], r["coords"])
            assert a["area"] == r["area"] and abs(a["perimeter"] - r["perimeter"]) <= 1e-12 * r["perimeter"]
            assert abs(a["cx"] - r["centroid_raw"][0]) < 1e-9 and abs(a["cy"] - r["centroid_raw"][1]) < 1e-9
            assert a["cls"] - 1 == r["class_int"]
    # offsets are an exclusive scan in (tile, record) order and the pool is densely used
    assert np.array_equal(cells["offset"], np.concatenate([[0], np.cumsum(cells["n_pts"])[:-1]]))
    assert len(xy) == int(cells["n_pts"].sum()) and n_seen == len(cells)
    assert (cells["valid"] == 1).mean() > 0.95


def test_device_polygons_multi_component_and_tiny(cuda):
    """hand-made id map: an instance with two components (the LAST one in raster order is contoured, like
    OpenCV's contours[0]), a single pixel, a thin line (invalid polygon), a U shape"""
    import ctypes as C
    from classpose_amd import _lib
    from classpose_amd._lib import ptr
    H = W = 64
    m = np.zeros((H, W), np.uint16)
    m[5:10, 5:10] = 1; m[20:24, 30:36] = 1                  # label 1: two components
    m[40, 40] = 2                                           # single pixel
    m[50, 10:20] = 3                                        # 1-px line: < 4 vertices or zero area
    m[10:20, 40:42] = 4; m[10:20, 48:50] = 4; m[18:20, 40:50] = 4    # U shape (two local tops, one component)
    L = _lib.lib()
    masks = torch.from_numpy(m.view(np.int16)).to(cuda)[None]
    cm = torch.ones((1, H, W), dtype=torch.uint8, device=cuda)
    max_rec = 16
    recs = torch.zeros(max_rec * C.sizeof(_lib.CpxRecord), dtype=torch.uint8, device=cuda)
    cnt = torch.zeros(1, dtype=torch.int32, device=cuda)
    ws = torch.empty(L.cpx_postproc_workspace_bytes(1, H, W), dtype=torch.uint8, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.cpx_instance_records(ptr(masks), ptr(cm), 1, H, W, max_rec, ptr(recs), ptr(cnt), ptr(ws), st))
    cells = torch.zeros(max_rec * C.sizeof(_lib.CpxCell), dtype=torch.uint8, device=cuda)
    pool = torch.zeros((512, 2), dtype=torch.float64, device=cuda)
    tot = torch.zeros(1, dtype=torch.int32, device=cuda)
    org = torch.zeros((1, 2), dtype=torch.float64, device=cuda)
    pws = torch.empty(L.cpx_polygonize_workspace_bytes(1, H, W, max_rec), dtype=torch.uint8, device=cuda)
    _lib.check(L.cpx_polygonize_device(ptr(masks), ptr(recs), ptr(cnt), 1, H, W, max_rec, 1.0, ptr(org), ptr(pool), 512,
                                       ptr(cells), ptr(tot), ptr(pws), st))
    n = int(cnt.item())
    assert n == 4
    dc = cells.cpu().numpy().view(engine.CELL_DTYPE)[:n]
    rr = recs.cpu().numpy().view(engine.RECORD_DTYPE)[:n]
    hc, hxy = postprocess.polygonize_tile(m, rr, 1.0, (0, 0))
    xy = pool.cpu().numpy()
    for name in ("area", "perimeter", "cx", "cy", "n_pts", "valid"):
        assert np.array_equal(dc[name], hc[name]), name
    for a, b in zip(dc, hc):
        assert np.array_equal(xy[a["offset"]: a["offset"] + a["n_pts"]], hxy[b["offset"]: b["offset"] + b["n_pts"]])
    by_label = {int(r["label"]): c for r, c in zip(rr, dc)}
    v1 = xy[by_label[1]["offset"]: by_label[1]["offset"] + by_label[1]["n_pts"]]
    assert v1.tolist() == [[30, 20], [30, 23], [35, 23], [35, 20]]            # the second component
    assert by_label[2]["n_pts"] == 1 and by_label[2]["valid"] == 0 and by_label[3]["valid"] == 0
    assert by_label[4]["valid"] == 1 and by_label[4]["n_pts"] >= 8
    # oracle: every label's contour is OpenCV-order contours[0] of its bbox crop; validity as GEOS decides it
    for lab, c in by_label.items():
        ys, xs = np.nonzero(m == lab)
        cont = opoly.find_contours_external_simple((m == lab)[ys.min(): ys.max() + 1, xs.min(): xs.max() + 1])[0]
        exp = (cont + [xs.min(), ys.min()]).astype(np.float64)
        assert np.array_equal(xy[c["offset"]: c["offset"] + c["n_pts"]], exp), lab
        assert bool(c["valid"]) == opoly.polygon_metrics(exp)["valid"], lab



def test_device_records_and_polygons_equal_reference_postprocessor(cuda):
    """cpx_instance_records + cpx_polygonize_device on the id / class maps of the reference's own ``eval`` run against the
    cells the reference's own ``PostProcessor.__call__`` produced from them (tests/golden/reference_postprocessor.npz):
    vertices, area, perimeter, rounded centroid, class -- per valid cell, in label order."""
    import ctypes as C
    import os
    from classpose_amd import _lib
    from classpose_amd._lib import ptr
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, ev = np.load(os.path.join(gd, "reference_postprocessor.npz")), np.load(os.path.join(gd, "reference_eval.npz"))
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for k in range(int(g["pp_n"])):
        case, cx, cy, labelled = (int(v) for v in g[f"pp_{k}_cfg"])
        scale = float(g[f"pp_{k}_scale"])
        m = ev[f"ev_{case}_masks"].astype(np.uint16).copy()
        cmh = ev[f"ev_{case}_class_masks"].astype(np.uint8).copy()
        if int(g[f"pp_{k}_extra"]):
            n = int(m.max())
            m[0, 0] = n + 1; m[5, 250:252] = n + 2
            cmh[0, 0] = 1; cmh[5, 250:252] = 2
        H, W = m.shape
        masks = torch.from_numpy(m.view(np.int16)).to(cuda)[None]
        cm = torch.from_numpy(cmh).to(cuda)[None]
        max_rec = 256
        recs = torch.zeros(max_rec * C.sizeof(_lib.CpxRecord), dtype=torch.uint8, device=cuda)
        cnt = torch.zeros(1, dtype=torch.int32, device=cuda)
        ws = torch.empty(L.cpx_postproc_workspace_bytes(1, H, W), dtype=torch.uint8, device=cuda)
        _lib.check(L.cpx_instance_records(ptr(masks), ptr(cm), 1, H, W, max_rec, ptr(recs), ptr(cnt), ptr(ws), st))
        cells = torch.zeros(max_rec * C.sizeof(_lib.CpxCell), dtype=torch.uint8, device=cuda)
        pool = torch.zeros((16384, 2), dtype=torch.float64, device=cuda)
        tot = torch.zeros(1, dtype=torch.int32, device=cuda)
        org = torch.tensor([[float(cx), float(cy)]], dtype=torch.float64, device=cuda)
        pws = torch.empty(L.cpx_polygonize_workspace_bytes(1, H, W, max_rec), dtype=torch.uint8, device=cuda)
        _lib.check(L.cpx_polygonize_device(ptr(masks), ptr(recs), ptr(cnt), 1, H, W, max_rec, scale, ptr(org), ptr(pool), 16384,
                                           ptr(cells), ptr(tot), ptr(pws), st))
        n = int(cnt.item())
        dc = cells.cpu().numpy().view(engine.CELL_DTYPE)[:n]
        xy = pool.cpu().numpy()
        dv = dc[dc["valid"] == 1]
        offs = np.concatenate([[0], np.cumsum(g[f"pp_{k}_n_pts"])])
        assert len(dv) == len(offs) - 1 and n - len(dv) == int(g[f"pp_{k}_n_invalid"])
        for i, c in enumerate(dv):
            want = g[f"pp_{k}_coords"][offs[i]: offs[i + 1]]
            assert np.array_equal(xy[c["offset"]: c["offset"] + c["n_pts"]], want[:-1])
            assert c["area"] == g[f"pp_{k}_area"][i] and abs(c["perimeter"] - g[f"pp_{k}_perimeter"][i]) <= 1e-12 * c["perimeter"]
            assert np.round([c["cx"], c["cy"]], 2).tolist() == g[f"pp_{k}_centroid"][i].tolist()
            if labelled:
                assert c["cls"] - 1 == int(g[f"pp_{k}_class_int"][i])
