"""-m "not gpu": the data-parallel path (objcavit_amd/dp.py) with world_size 2 on
the gloo backend: contiguous sharding by rank, per-image metric records, ONE
all-gather, and the gathered table == the single-process table, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from objcavit_amd import dp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_depths(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(n, 1, 24, 32, generator=g) * 9.5 + 0.2
    gt[:, :, :2, :] = 0.0                                   # invalid pixels (no ground truth)
    pred = gt * (1 + 0.2 * (torch.rand(n, 1, 24, 32, generator=g) - 0.5)) + 0.01
    return pred, gt


def _worker(rank, world, port, n_images, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = dp.init_from_env("cpu")
    assert (r, w) == (rank, world)
    pred, gt = _fake_depths(n_images)
    lo, hi = dp.shard_range(n_images, rank, world)
    rec = dp.per_image_metrics(pred[lo:hi], gt[lo:hi], 0.001, 10.0, first_image_id=lo)
    table = dp.gather_records(rec, world)
    if rank == 0:
        torch.save(table, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_per_image_metrics_formulas():
    """AbsRel = mean(|gt - pred| / gt) over valid pixels (reference metrics/AbsRel.py:23); delta thresholds
    (reference metrics/AccThresh.py:31-32)."""
    pred, gt = _fake_depths(3)
    rec = dp.per_image_metrics(pred, gt, 0.001, 10.0)
    for i in range(3):
        m = (gt[i] > 0.001) & (gt[i] < 10.0)
        p, g = pred[i].clamp(0.001, 10.0)[m].double(), gt[i][m].double()
        assert abs(float(rec[i, 0]) - float(((g - p).abs() / g).mean())) < 1e-6
        assert abs(float(rec[i, 2]) - float(((g - p) ** 2).mean().sqrt())) < 1e-5
        ratio = torch.maximum(g / p, p / g)
        assert abs(float(rec[i, 5]) - float((ratio < 1.25).double().mean())) < 1e-6
        assert int(rec[i, 8]) == int(m.sum()) and int(rec[i, 9]) == i


@pytest.mark.timeout(120)
def test_two_rank_gather_equals_single_process(tmp_path):
    n, world = 8, 2
    out = str(tmp_path / "table.pt")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    table = torch.load(out)
    pred, gt = _fake_depths(n)
    single = dp.per_image_metrics(pred, gt, 0.001, 10.0)
    assert table.shape == single.shape
    assert torch.equal(table, single)                       # sharding + one all-gather changes nothing
    assert table[:, 9].tolist() == list(range(n))           # rank-ordered, contiguous image ids
    s = dp.summarise(table)
    assert s["images"] == n and 0 < s["abs_rel"] < 0.2
