"""GPU: the N > 1 path over RCCL (backend "nccl") with one fresh process per GPU.  Runs when the box has
>= 2 GPUs and skips otherwise, like /root/reference/tests/test_prediction_integration.py:131-133."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from classpose_amd import engine, parallel, synth          # parallel sets HSA_ENABLE_IPC_MODE_LEGACY=0 (RCCL needs dmabuf IPC here)
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    if world == 1:           # init_distributed is a no-op for one rank: create the 1-rank RCCL communicator explicitly
        torch.cuda.set_device(0)
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        r, w, local = 0, 1, 0
    else:
        r, w, local = parallel.init_distributed("nccl")
    dev = torch.device("cuda", local)
    # one engine step on this rank's shard of a 3 x 3 tile grid, then the path's only collective
    sd = synth.make_state_dict(7, None, depth=1, seed=0)
    eng = engine.Engine(engine.NetWeights.from_state_dict(sd, "bf16", dev), 256, batch_tiles=9 if w == 1 else 5)
    coords = [(224 * i, 224 * j) for i in range(3) for j in range(3)]
    mine = list(parallel.shard_indices(len(coords), r, w))
    tiles = np.stack([synth.render_region(7, x, y, 256, 256) for x, y in (coords[k] for k in mine)])
    inj = [synth.analytic_fields(7, *coords[k], 256, 256, 7) for k in mine]
    inject = tuple(torch.from_numpy(np.stack([a[c] for a in inj])).to(dev) for c in range(3))
    out = eng.run(torch.from_numpy(tiles).to(dev), inject=inject)
    rec = eng.fetch_records(len(mine), out)
    rec["tile"] = np.asarray(mine)[rec["tile"]]                   # global tile index
    t = torch.from_numpy(rec.view(np.uint8).reshape(len(rec), -1).copy()).to(dev)
    allrec = parallel.all_gather_records(t).cpu().numpy().reshape(-1).view(engine.RECORD_DTYPE)
    empty = parallel.all_gather_records(torch.zeros((0 if r else 3, 48), dtype=torch.uint8, device=dev)).cpu().numpy()
    mx = parallel.allreduce_max(float(len(rec)), dev)
    root = parallel.gather_to_root(t)
    parallel.barrier()
    with open("/proc/self/maps") as f:
        rccl = any("librccl" in line or "libnccl" in line for line in f)
    q.put((r, len(rec), allrec.tobytes(), empty.shape, mx, torch.distributed.get_backend(), rccl,
           None if root is None else tuple(root.shape)))
    torch.distributed.destroy_process_group()


def test_two_ranks_nccl_shard_and_gather():
    if torch.cuda.device_count() < 2:
        pytest.skip("Needs at least 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=600) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    from classpose_amd import engine
    a = np.frombuffer(res[0][2], dtype=engine.RECORD_DTYPE)
    b = np.frombuffer(res[1][2], dtype=engine.RECORD_DTYPE)
    assert np.array_equal(a, b) and len(a) == res[0][1] + res[1][1] > 200
    assert set(a["tile"][:res[0][1]]) == {0, 2, 4, 6, 8} and set(a["tile"][res[0][1]:]) == {1, 3, 5, 7}   # rank order
    assert res[0][3] == res[1][3] == (3, 48) and res[0][4] == res[1][4] == max(res[0][1], res[1][1])


def test_one_rank_nccl_collectives_run_on_rccl():
    """What a 1-GPU box can show of the RCCL path: a ONE-rank "nccl" process group (communicator creation under
    HSA_ENABLE_IPC_MODE_LEGACY=0, device tensors through ncclAllGather / ncclAllReduce, the barrier) runs the same exchange
    code as N ranks -- counts + padded all-gather of the 48-byte rows, max-reduce, gather-to-root -- with librccl mapped."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), q))
    p.start()
    r, n, raw, empty_shape, mx, backend, rccl, root_shape = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    from classpose_amd import engine
    a = np.frombuffer(raw, dtype=engine.RECORD_DTYPE)
    assert backend == "nccl" and rccl, (backend, rccl)
    assert len(a) == n > 400 and set(a["tile"]) == set(range(9))
    assert empty_shape == (3, 48) and mx == float(n) and root_shape == (n, a.dtype.itemsize)
