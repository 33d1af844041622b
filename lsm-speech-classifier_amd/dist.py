"""Clip sharding across the GPUs of one node + the one exchange step of the path.

The reference is a single serial loop over clips (/root/reference/extract_lsm_features.py:78);
clips are independent (``lsm.reset()`` precedes each one, :79), so they shard with no data-path
collective: rank r takes the contiguous block [lo, hi) of the dataset (gather order = dataset
order) and the fp32 feature rows are all-gathered once at the end (RCCL over xGMI through
``torch.distributed``; ``gloo`` on CPU in the tests).  StandardScaler and the npz write stay on
rank 0 so that results equal the single-process run exactly.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process when absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str | None = None):
    """Join the process group when launched under torchrun; no-op for a single process.
    Rehearsal switches for a one-GPU box (never needed on a real node): ``LSM_DIST_BACKEND=gloo`` exchanges
    through the host, ``LSM_SHARE_GPU=1`` puts every rank on cuda:0."""
    rank, local_rank, world = env_world()
    if os.environ.get("LSM_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("LSM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            # a host backend does not need one GPU per rank: take this rank's GPU when the box has it, else share cuda:0
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank if local_rank < torch.cuda.device_count() else 0)
            dist.init_process_group(backend)
    return rank, local_rank, world


def local_device() -> torch.device:
    """The GPU of this rank (LOCAL_RANK; cuda:0 for a single process or under LSM_SHARE_GPU=1)."""
    _, local_rank, world = env_world()
    if world <= 1:
        return torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    if os.environ.get("LSM_SHARE_GPU") == "1" or (dist.is_initialized() and dist.get_backend() != "nccl"
                                                  and local_rank >= torch.cuda.device_count()):
        local_rank = 0
    return torch.device("cuda", local_rank)


def group_world():
    """(rank, world_size) of the INITIALISED process group; (0, 1) when there is none -- so a caller that did
    not go through init() never shards by a WORLD_SIZE it cannot gather over.  Raises when the launcher
    environment announces several ranks but no group was initialised (the rows would silently be a shard)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: call dist.init() first")
    return 0, 1


def finish():
    """Leave the process group (end of main())."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def shard_bounds(n: int, world: int):
    """Per-rank [lo, hi): contiguous blocks of ceil(n/world) clips (the last ones may be short or
    empty)."""
    per = -(-n // world) if world > 0 else n
    return [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]


def shard_range(n: int, rank: int, world: int):
    return shard_bounds(n, world)[rank]


def gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather the per-rank row blocks of ``shard_bounds`` back into dataset order.
    ``local`` is (hi-lo, F) on this rank's device; every rank returns (n_total, F)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = -(-n_total // world)
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return out[:n_total]


def gather_varrows(local: torch.Tensor) -> torch.Tensor:
    """All-gather row blocks whose lengths the ranks do not know of each other (stage 1: a rank skips the wav
    files it cannot read), in rank order.  ``local`` is (n_r, F) on this rank's device, F equal on every rank;
    every rank returns (sum n_r, F).  Two collectives: the counts, then the blocks padded to the longest."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    counts = torch.zeros((world,), dtype=torch.int64, device=local.device)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, mine)
    counts = [int(c) for c in counts.cpu()]
    per = max(counts) if counts else 0
    if per == 0:
        return local[:0]
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * per: r * per + counts[r]] for r in range(world)])


def broadcast_float(value: float, src: int = 0, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.broadcast(t, src)
    return float(t.item())


# ---- the rows of a run of steps, exchanged in chunks (bench.py --exchange chunked) ------------------------------------
# Every rank keeps (n_steps, B, F) rows; the rows of steps [c0, c1) travel in one all-gather as soon as those steps have
# finished, into a block laid out chunk by chunk, each chunk rank by rank.  One place for that layout: the bench writes
# with `gather_step_chunk` and reads with `rows_by_rank`; tests/test_dist_gloo.py runs both over gloo on the CPU.

def chunk_bounds(n_steps: int, chunk: int):
    """[c0, c1) of every chunk of `chunk` steps (the last one may be short)."""
    chunk = max(1, int(chunk))
    return [(c0, min(c0 + chunk, n_steps)) for c0 in range(0, n_steps, chunk)]


def gather_step_chunk(gathered: torch.Tensor, local_rows: torch.Tensor, c0: int, c1: int) -> None:
    """All-gather the rows of steps [c0, c1) of every rank.  `local_rows` is (n_steps, B, F) on this rank, `gathered`
    (world * n_steps * B, F): the chunk occupies rows [world*c0*B, world*c1*B), rank r's steps inside it follow one
    another.  Runs on the current stream (the caller orders it behind the chunk's steps)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    B, F = local_rows.shape[1], local_rows.shape[2]
    src = local_rows[c0:c1].reshape((c1 - c0) * B, F)
    dst = gathered[world * c0 * B: world * c1 * B]
    if dist.is_initialized():            # also with one rank: the rehearsal on a 1-GPU box goes through RCCL's call path
        dist.all_gather_into_tensor(dst, src)
    else:
        dst.copy_(src)


def rows_by_rank(gathered: torch.Tensor, world: int, n_steps: int, chunk: int, rows_per_step: int):
    """The gathered block back in (rank, step, clip) order: a list of `world` tensors (n_steps * rows_per_step, F)."""
    B = rows_per_step
    return [torch.cat([gathered[world * c0 * B + r * (c1 - c0) * B: world * c0 * B + (r + 1) * (c1 - c0) * B]
                       for c0, c1 in chunk_bounds(n_steps, chunk)]) for r in range(world)]


def rows_digest(blocks) -> int:
    """Order-sensitive 64-bit digest of row blocks (float32 bit patterns weighted by position, wrapping sum): equal for
    two exchanges that delivered the same rows in the same (rank, step, clip) order."""
    canon = torch.cat(list(blocks)).contiguous().view(torch.int32).to(torch.int64).reshape(-1)
    weights = torch.arange(canon.numel(), device=canon.device, dtype=torch.int64) % 65521 + 1
    return int((canon * weights).sum().item())

