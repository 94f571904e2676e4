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
from typing import Dict, List, Optional, Sequence, Tuple

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


def agree_object_nmax(local_counts: Sequence[int], world: int, device=None, global_counts: Optional[Sequence[int]] = None) -> int:
    """The longest object list of the GLOBAL batch, identical on every rank.

    SURVEY.md Q3: with ``use_2_saca`` the reference pads every object list to the batch's longest one (modules/
    ObjCAViT.py:180-183) and the second SA/CA stack sees the padding rows, so an image's result depends on the Nmax of
    the batch it sits in.  A shard that pads only to ITS longest list therefore differs from the single-process run in
    the fifth digit whenever the ranks' maxima differ.  ``global_counts`` (every image's count, known on the host --
    e.g. the benchmark's fixed counts or a detection cache): plain arithmetic, no collective.  Otherwise one MAX
    all-reduce of a single integer before the forward."""
    if global_counts is not None:
        nmax = max(int(c) for c in global_counts)
        if local_counts and max(int(c) for c in local_counts) > nmax:
            raise ValueError("agree_object_nmax: a local count exceeds the largest of global_counts")
        return nmax
    local = max((int(c) for c in local_counts), default=0)
    if world == 1:
        return local
    t = torch.tensor([local], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def sharded_forward(model, image: torch.Tensor, object_features: List[torch.Tensor], object_xywh_list: List[Optional[torch.Tensor]],
                    world: int, global_counts: Optional[Sequence[int]] = None):
    """One rank's forward over its shard of a global batch, reproducing the single-process result image for image: when
    the model's second SA/CA stack is on (``use_2_saca``) the object lists are padded to the global Nmax
    (``agree_object_nmax``); otherwise an image's result does not depend on its batch and nothing is exchanged."""
    objcavit = getattr(model, "objcavit", None)
    if objcavit is None or not getattr(objcavit, "use_2_saca", False):
        return model(image, object_features, object_xywh_list)
    counts = [int(f.shape[0]) for f in object_features]
    dev = image.device if image.device.type == "cuda" else None
    nmax = agree_object_nmax(counts, world, dev, global_counts)
    return model(image, object_features, object_xywh_list, pad_objects_to=nmax)


def summarise(records: torch.Tensor) -> Dict[str, float]:
    """Image-wise averages (the reference's ``*RunningAvg`` metrics) from the gathered table."""
    r = drop_padding(records).double().cpu()
    return {name: float(r[:, i].mean()) for i, name in enumerate(RECORD_FIELDS[:8])} | {"images": int(r.shape[0])}
