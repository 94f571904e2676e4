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
from typing import Dict, List, Tuple

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


def per_image_metrics(pred: torch.Tensor, gt: torch.Tensor, min_depth: float, max_depth: float,
                      first_image_id: int = 0) -> torch.Tensor:
    """B x len(RECORD_FIELDS) fp32 record per image (valid = gt inside (min_depth, max_depth)).
    pred / gt: B x 1 x h x w at the same resolution."""
    B = pred.shape[0]
    p = pred.reshape(B, -1).clamp(min_depth, max_depth)
    g = gt.reshape(B, -1)
    valid = (g > min_depth) & (g < max_depth)
    n = valid.sum(1).clamp(min=1).to(p.dtype)
    gs = torch.where(valid, g, torch.ones_like(g))
    ps = torch.where(valid, p, torch.ones_like(p))
    vf = valid.to(p.dtype)

    def mean(t):
        return (t * vf).sum(1) / n

    ratio = torch.maximum(gs / ps, ps / gs)
    rec = torch.stack([
        mean((gs - ps).abs() / gs),
        mean((gs - ps) ** 2 / gs),
        torch.sqrt(mean((gs - ps) ** 2)),
        torch.sqrt(mean((torch.log(gs) - torch.log(ps)) ** 2)),
        mean((torch.log10(gs) - torch.log10(ps)).abs()),
        mean((ratio < 1.25).to(p.dtype)), mean((ratio < 1.25 ** 2).to(p.dtype)), mean((ratio < 1.25 ** 3).to(p.dtype)),
        valid.sum(1).to(p.dtype),
        torch.arange(first_image_id, first_image_id + B, device=p.device, dtype=p.dtype),
    ], dim=1)
    return rec.float().contiguous()


def gather_records(local: torch.Tensor, world: int) -> torch.Tensor:
    """The single collective of an evaluation step: all-gather of the packed per-image records
    (equal shard sizes) -> (world * B_local) x F, ordered by rank."""
    if world == 1:
        return local
    out = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def summarise(records: torch.Tensor) -> Dict[str, float]:
    """Image-wise averages (the reference's ``*RunningAvg`` metrics) from the gathered table."""
    r = records.double().cpu()
    return {name: float(r[:, i].mean()) for i, name in enumerate(RECORD_FIELDS[:8])} | {"images": int(r.shape[0])}
