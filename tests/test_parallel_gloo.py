"""CPU: the N>1 path (tile sharding + record all-gather) over gloo, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from classpose_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed("gloo")
    idx = list(parallel.shard_indices(11, r, w))
    # each rank "finds" a different number of cells; rank 1 may even find none
    n = 5 if r == 0 else 0
    rec = torch.full((n, 48), r + 1, dtype=torch.uint8)
    rec[:, 0] = torch.arange(n, dtype=torch.uint8)
    allrec = parallel.all_gather_records(rec)
    rec2 = torch.full((3 + r, 48), 10 + r, dtype=torch.uint8)
    allrec2 = parallel.all_gather_records(rec2)
    counts = parallel.all_gather_counts(3 + r, torch.device("cpu"))
    pool = torch.full((2 + 3 * r, 16), 20 + r, dtype=torch.uint8)       # "vertex pool": only the writing rank receives it
    rooted = parallel.gather_to_root(pool, 0)
    assert counts == [3 + i for i in range(w)]
    if r == 0:
        assert rooted.shape == (sum(2 + 3 * i for i in range(w)), 16)
        o = 0
        for i in range(w):
            assert bool((rooted[o:o + 2 + 3 * i] == 20 + i).all())
            o += 2 + 3 * i
    else:
        assert rooted is None
    empty = parallel.gather_to_root(torch.zeros((0, 16), dtype=torch.uint8) if r else torch.ones((1, 16), dtype=torch.uint8), 0)
    assert (empty.shape == (1, 16)) if r == 0 else (empty is None)
    mx = parallel.allreduce_max(float(r + 1), torch.device("cpu"))
    sm = parallel.allreduce_sum(float(r + 1), torch.device("cpu"))
    parallel.barrier()
    q.put((r, idx, allrec.numpy(), allrec2.numpy(), mx, sm))
    torch.distributed.destroy_process_group()


def test_shard_and_all_gather_records_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6, 8, 10] and res[1][1] == [1, 3, 5, 7, 9]
    for r in res:
        assert r[2].shape == (5, 48) and np.all(r[2][:, 1] == 1)          # only rank 0 had records
        assert r[3].shape == (7, 48)
        assert np.all(r[3][:3, 5] == 10) and np.all(r[3][3:, 5] == 11)   # rank order preserved
        assert r[4] == 2.0 and r[5] == 3.0
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


def test_gather_to_root_world4():
    """four ranks: rows all-gathered, pools gathered to rank 0 only (incl. empty contributions)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[0, 4, 8], [1, 5, 9], [2, 6, 10], [3, 7]]
    for r in res:
        assert r[3].shape == (3 + 4 + 5 + 6, 48) and r[4] == 4.0 and r[5] == 10.0


def _cells_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    from classpose_amd.entrypoints import predict_wsi as pw
    n = 3 + 2 * rank                                   # cells of this rank, 4 vertices each
    cells = np.zeros(n, pw.CELL_ROW)
    cells["area"] = 100 * rank + np.arange(n); cells["n_pts"] = 4; cells["cls"] = rank + 1
    xy = (np.arange(n * 4 * 2, dtype=np.float64).reshape(-1, 2) + 1000 * rank)
    tiles = np.arange(n, dtype=np.int64) * world + rank
    c, v, t = pw.gather_cells(cells, xy, torch.device("cpu"), tiles)
    parallel.barrier()
    q.put((rank, c.copy(), None if v is None else v.copy(), t.copy()))
    torch.distributed.destroy_process_group()


def test_gather_cells_rows_everywhere_vertices_on_rank0_only():
    """the CLI's exchange (predict_wsi.gather_cells) over gloo, world 3: cell rows + tile indices on every rank in rank order,
    the vertex pool on rank 0 only"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cells_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    n_tot = 3 + 5 + 7
    for r, c, v, t in res:
        assert len(c) == n_tot and len(t) == n_tot
        assert list(c["cls"]) == [1] * 3 + [2] * 5 + [3] * 7
        assert np.array_equal(c.view(np.uint8), res[0][1].view(np.uint8)) and np.array_equal(t, res[0][3])
        if r == 0:
            assert v.shape == (n_tot * 4, 2) and v[0, 0] == 0 and v[12, 0] == 1000 and v[12 + 20, 0] == 2000
        else:
            assert v is None


def test_single_process_passthrough():
    rec = torch.zeros((4, 48), dtype=torch.uint8)
    assert parallel.all_gather_records(rec) is rec
    assert list(parallel.shard_indices(5, 0, 1)) == [0, 1, 2, 3, 4]
