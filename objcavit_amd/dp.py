"""Data-parallel inference over the GPUs of one node: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on
CPU for tests), images sharded contiguously by rank, weights replicated, NO
exchange during the forward, and ONE all-gather of a packed per-image metric
record at the end.  The reference reaches the same numbers
through 32 scalar torchmetrics gathers (metrics/*.py ``add_state(...,
dist_reduce_fx=...)``; formulas metrics/AbsRel.py:23 etc.).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

RECORD_FIELDS = ("abs_rel", "sq_rel", "rmse", "rmse_log", "log10", "delta1", "delta2", "delta3", "n_valid", "image_id")


def init_from_env(device_type: str = "cuda") -> Tuple[int, int, int]:
    """(rank, local_rank, world).  Initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl" if device_type == "cuda" else "gloo", rank=rank, world_size=world)
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rows_per_rank(n_items: int, world: int) -> int:
    """Rows every rank contributes to the gather: the largest shard of ``shard_range`` (shards differ by at most one)."""
    return -(-n_items // world) if n_items > 0 else 0


def pad_records(local: torch.Tensor, rows: int) -> torch.Tensor:
    """Pad a rank's record table to ``rows`` rows with empty records (n_valid = 0, image_id = -1)."""
    if local.shape[0] > rows:
        raise ValueError(f"rank holds {local.shape[0]} records, more than the agreed {rows} per rank")
    if local.shape[0] == rows:
        return local.contiguous()
    pad = torch.zeros(rows - local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    pad[:, RECORD_FIELDS.index("image_id")] = -1.0
    return torch.cat([local, pad], 0)


def drop_padding(table: torch.Tensor) -> torch.Tensor:
    return table[table[:, RECORD_FIELDS.index("image_id")] >= 0]


def gather_records(local: torch.Tensor, world: int, n_total: Optional[int] = None) -> torch.Tensor:
    """The single collective of an evaluation job: all-gather of the packed per-image records -> table ordered by
    rank (= by image id for contiguous shards).

    ``all_gather_into_tensor`` needs the same row count on every rank, and ``shard_range`` hands out shards that
    differ by one when ``n_total`` is not a multiple of ``world`` (654 NYU / 697 KITTI evaluation images on 8 ranks):
    with ``n_total`` every rank pads its table to ``rows_per_rank(n_total, world)`` rows with empty records
    (image_id = -1) that are dropped again after the gather -- the row count is agreed from host-side arithmetic, so
    this stays ONE collective.  Without ``n_total`` the caller asserts equal shards; that is verified with a MIN/MAX
    all-reduce of the row counts (a mismatch would otherwise hang RCCL or return garbage rows)."""
    if world == 1:
        return local
    if n_total is not None:
        local = pad_records(local, rows_per_rank(n_total, world))
    else:
        n = torch.tensor([local.shape[0], -local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        if int(n[0]) != -int(n[1]):
            raise RuntimeError(f"gather_records: ranks hold between {-int(n[1])} and {int(n[0])} records; pass n_total "
                               "so that every rank pads to the same row count")
    out = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return drop_padding(out) if n_total is not None else out


def summarise(records: torch.Tensor) -> Dict[str, float]:
    """Image-wise averages (the reference's ``*RunningAvg`` metrics) from the gathered table."""
    r = drop_padding(records).double().cpu()
    return {name: float(r[:, i].mean()) for i, name in enumerate(RECORD_FIELDS[:8])} | {"images": int(r.shape[0])}
