"""Multi-GPU plumbing for the tile path: static tile sharding + one record all-gather.

The reference shares ONE multiprocessing queue between per-device worker processes
and funnels every (masks, class_masks) array through one PostProcessor process
(/root/reference/src/classpose/entrypoints/predict_wsi.py:1542-1572, 547-566);
it uses no collective in inference.  Here each rank (one process per GPU, launched
by torch.distributed.run) owns the tiles ``k % world == rank`` of the
``_get_coords`` order -- no data-path collective -- and the only exchange is a
variable-length all-gather of the compact per-cell records at the end of the
slide (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# The pool's host driver only supports dmabuf IPC: without this setting RCCL's intra-node transport (and any sharing of
# device tensors across processes) fails with ``hipIpcGetMemHandle: invalid argument``.  It has to be in the environment
# before the HIP runtime initialises, so it is set when this module is imported (every multi-process entry point --
# bench.py, the CLI's per-GPU children, the tests -- imports it first); an explicit value in the environment wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def env_rank_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for world 1)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # the ranks cannot each pick a free port, and a fixed default (29500 until round 4) makes two jobs on one node collide:
            # whoever starts the ranks chooses the port (torch.distributed.run; the CLI's --device cuda:0,1 parent and bench.py's
            # --gpus N parent probe a free one and hand it to their children)
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: start the ranks with torch.distributed.run, "
                               "`classpose-predict-wsi --device cuda:0,1,...` or `bench.py --gpus N`, which choose a free port")
        if backend is None:
            backend = os.environ.get("CPX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> range:
    """Tile k of the x-major ``_get_coords`` order goes to rank k % world."""
    return range(rank, n_items, world)


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def _coll_device(device):
    """gloo (CPU tests / single-GPU dry runs) moves collectives through host tensors."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def allreduce_max(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_sum(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_gather_counts(n_local: int, device) -> list[int]:
    """How many records every rank contributes (one tiny all-gather; [n_local] without a process group)."""
    if not dist.is_initialized():
        return [int(n_local)]
    dev = _coll_device(device)
    n = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    counts = torch.empty(dist.get_world_size(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n)
    return [int(c) for c in counts.tolist()]


def gather_to_root(rec: torch.Tensor, root: int = 0) -> torch.Tensor | None:
    """Variable-length GATHER of rows to one rank (point-to-point sends; nothing is padded, nothing reaches the other
    ranks): what only the writing rank needs -- the polygon vertex pools, ~0.5 KB per cell, 5-6 GB on the 80 000^2 TTA
    slide -- must not be all-gathered (round-3 review).  rec: [n_local, width] uint8 on this rank's device.  Returns
    [sum n, width] in rank order on ``root`` (on rec's device), None elsewhere."""
    if not dist.is_initialized():
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    out_dev = rec.device
    dev = _coll_device(rec.device)
    counts = all_gather_counts(rec.shape[0], rec.device)
    width = rec.shape[1]
    if rank != root:
        if counts[rank]:
            dist.send(rec.to(dev).contiguous(), dst=root)
        return None
    parts = []
    for r in range(world):
        if r == root:
            parts.append(rec.to(dev))
        elif counts[r]:
            buf = torch.empty((counts[r], width), dtype=torch.uint8, device=dev)
            dist.recv(buf, src=r)
            parts.append(buf)
    return torch.cat(parts, 0).to(out_dev) if parts else rec


def all_gather_records(rec: torch.Tensor) -> torch.Tensor:
    """Variable-length all-gather of per-cell records.

    rec: [n_local, rec_bytes] uint8 on this rank's device (n_local may be 0).
    Returns [sum n, rec_bytes] with rank 0's records first (deterministic order).
    Implemented as an all-gather of counts followed by one padded
    all_gather_into_tensor: the payload is tiny (48 B per cell), so a single
    latency-bound exchange beats anything cleverer on point-to-point xGMI.
    """
    if not dist.is_initialized():
        return rec
    world = dist.get_world_size()
    out_dev = rec.device
    dev = _coll_device(rec.device)
    rec = rec.to(dev)
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n)
    mx = max(int(counts.max().item()), 1)
    width = rec.shape[1]
    padded = torch.zeros((mx, width), dtype=torch.uint8, device=dev)
    padded[: rec.shape[0]] = rec
    out = torch.empty((world * mx, width), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, padded)
    out = out.view(world, mx, width)
    return torch.cat([out[r, : int(counts[r])] for r in range(world)], 0).to(out_dev)
