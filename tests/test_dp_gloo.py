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
from oracle import validation_ref


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_depths(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(n, 1, 24, 32, generator=g) * 9.5 + 0.2
    gt[:, :, :2, :] = 0.0                                   # invalid pixels (no ground truth)
    gt[:, :, 2, :] = 10.0                                   # exactly max_depth: VALID (metrics/MetricsPreprocess.py:26, <=)
    pred = gt * (1 + 0.2 * (torch.rand(n, 1, 24, 32, generator=g) - 0.5)) + 0.01
    return pred, gt


def _records(pred, gt, first_image_id=0):
    """Per-image records as the device kernel produces them, here from the pinned CPU oracle of the validation step
    (the product has exactly one implementation of these formulas, csrc/metrics.hip; no GPU in this suite)."""
    return validation_ref.per_image_records(pred, gt, 0.001, 10.0, first_image_id=first_image_id)


def _worker(rank, world, port, n_images, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = dp.init_from_env("cpu")
    assert (r, w) == (rank, world)
    pred, gt = _fake_depths(n_images)
    lo, hi = dp.shard_range(n_images, rank, world)
    rec = _records(pred[lo:hi], gt[lo:hi], first_image_id=lo)
    assert rec.shape[0] == hi - lo
    table = dp.gather_records(rec, world, n_total=n_images)
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


def test_pad_and_drop_records():
    rec = _records(*_fake_depths(3))
    padded = dp.pad_records(rec, 5)
    assert padded.shape == (5, 10) and padded[3:, 9].tolist() == [-1.0, -1.0] and float(padded[3:, 8].sum()) == 0
    assert torch.equal(dp.drop_padding(padded), rec)
    assert dp.summarise(padded) == dp.summarise(rec)
    assert dp.rows_per_rank(654, 8) == 82 and dp.rows_per_rank(697, 8) == 88 and dp.rows_per_rank(16, 8) == 2
    with pytest.raises(ValueError):
        dp.pad_records(rec, 2)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("n", [8, 7])
def test_two_rank_gather_equals_single_process(tmp_path, n):
    """n = 7: shards of 4 and 3 images -- the short rank pads one empty record, the gathered table drops it."""
    world = 2
    out = str(tmp_path / "table.pt")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    table = torch.load(out)
    pred, gt = _fake_depths(n)
    single = _records(pred, gt)
    assert table.shape == single.shape
    assert torch.equal(table, single)                       # sharding + one all-gather changes nothing
    assert table[:, 9].tolist() == list(range(n))           # rank-ordered, contiguous image ids
    s = dp.summarise(table)
    assert s["images"] == n and 0 < s["abs_rel"] < 0.2


# ---------------------------------------------------------------------------
# SURVEY.md Q3 under sharding: with use_2_saca an image's result depends on the batch's longest object list.
# ---------------------------------------------------------------------------
NMAX_COUNTS = [70, 12, 5, 9]          # rank 0 holds [70, 12], rank 1 holds [5, 9]: the ranks' own maxima differ (70 vs 9)


def _nmax_case():
    """ObjCAViT (learned_bbox_wh, use_2_saca) weights and a 4-image batch on the mini 176 x 192 feature map (S = 132)."""
    import gen
    from util import gains_of, load_golden, state_dict_from
    meta, _ = load_golden("g3_objcavit_bbox_wh_2saca_many")
    sd = state_dict_from(meta["shapes"], meta["seed"], gains_of(meta))
    fh, fw, seed = meta["fh"], meta["fw"], 77
    x = gen.randn("x", (len(NMAX_COUNTS), 128, fh, fw), seed)
    feats = [gen.randn(f"f{i}", (n, 512), seed, 10.0 / np.sqrt(512)) for i, n in enumerate(NMAX_COUNTS)]
    xywh = [gen.boxes(f"b{i}", n, seed, 2 * fh, 2 * fw) for i, n in enumerate(NMAX_COUNTS)]
    return sd, meta["kw"], x, feats, xywh


def _nmax_worker(rank, world, port, out_path, agree):
    from oracle import restate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dp.init_from_env("cpu")
    torch.set_grad_enabled(False)
    sd, kw, x, feats, xywh = _nmax_case()
    lo, hi = dp.shard_range(len(NMAX_COUNTS), rank, world)
    counts = NMAX_COUNTS[lo:hi]
    nmax = dp.agree_object_nmax(counts, world) if agree else None       # ONE integer MAX all-reduce, before the forward
    if agree:
        assert nmax == max(NMAX_COUNTS)
        assert dp.agree_object_nmax(counts, world, global_counts=NMAX_COUNTS) == nmax      # host-side counts: no collective
    y, _, inter = restate.objcavit_forward(x[lo:hi], feats[lo:hi], xywh[lo:hi], sd, "", batch_nmax=nmax, return_intermediates=True, **kw)
    rec = torch.cat([y, inter["saca2_img"][:, :4].flatten(1), torch.arange(lo, hi, dtype=torch.float32)[:, None]], 1).contiguous()
    # = bin widths, the first four tokens leaving the second SA/CA stack, image id
    out = torch.empty(world * rec.shape[0], rec.shape[1])
    dist.all_gather_into_tensor(out, rec)
    if rank == 0:
        torch.save(out, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_with_different_nmax_reproduce_the_single_process_batch(tmp_path):
    """Rank 0 and rank 1 hold shards whose longest object lists differ.  Padding every shard to the agreed GLOBAL Nmax
    reproduces the single-process batch; padding to the shard's own maximum (what a naive shard does) does not."""
    from oracle import restate
    torch.set_grad_enabled(False)
    sd, kw, x, feats, xywh = _nmax_case()
    y, _, inter = restate.objcavit_forward(x, feats, xywh, sd, "", return_intermediates=True, **kw)
    single = torch.cat([y, inter["saca2_img"][:, :4].flatten(1)], 1)
    tables = {}
    for agree in (True, False):
        out = str(tmp_path / f"t{int(agree)}.pt")
        mp.spawn(_nmax_worker, args=(2, _free_port(), out, agree), nprocs=2, join=True)
        tables[agree] = torch.load(out)
    assert tables[True][:, -1].tolist() == [0.0, 1.0, 2.0, 3.0]
    dev_agree = float((tables[True][:, :-1] - single).abs().max() / single.abs().max())
    dev_naive = float((tables[False][2:, :-1] - single[2:]).abs().max() / single.abs().max())
    assert dev_agree <= 1e-7, dev_agree                # same arithmetic on the same padded rows (0.0 here)
    assert dev_naive > 5e-7 and dev_naive > 5 * dev_agree, (dev_naive, dev_agree)      # Q3: the coupling is real and is what the agreement removes
    assert float((tables[False][:2, :-1] - single[:2]).abs().max() / single.abs().max()) < 2e-6    # rank 0 held the global maximum anyway
