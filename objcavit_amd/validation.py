"""The reference's validation step around the model (row N2 of SURVEY.md section 8f), on the device.

``ValidationStep(model, args)(image, depth_gt)`` does what ``GraphBinsLM.validation_step`` does between the batch and
its logged numbers (modules/GraphBinsLM.py:154-212): forward on the image and on its mirror -- here as ONE forward over the
2B images [batch | mirrored batch] (the reference forces bs 1, main.py:58, and calls the model twice, :159,173: at that size
every launch is latency, so two calls cost twice one; images are independent and the one coupling between them, the Nmax the
object rows are padded to with ``use_2_saca``, is formed per GROUP of B images on the device) --, clamp, un-flip, average,
then metrics/MetricsPreprocess.py (resize to the ground truth, nan/inf fix, validity mask, Garg / Eigen crop) and the
eight metrics -- the last three steps in ONE kernel (csrc/metrics.hip) that returns one record per image.  A
data-parallel job all-gathers the records once (objcavit_amd/dp.py); ``dp.summarise`` gives the reference's
running-average numbers, ``totals`` below its pixel-total numbers.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import hip_ops
from .dp import RECORD_FIELDS


def crop_box(args, H: int, W: int) -> Optional[Tuple[int, int, int, int]]:
    """(y0, y1, x0, x1) evaluation box of metrics/MetricsPreprocess.py:28-43 for ``args.basic.dataset``, or None."""
    ds = args[args.basic.dataset]
    if ds.get("garg_crop", False):
        return int(0.40810811 * H), int(0.99189189 * H), int(0.03594771 * W), int(0.96405229 * W)
    if ds.get("eigen_crop", False):
        if args.basic.dataset == "kitti":
            return int(0.3324324 * H), int(0.91351351 * H), int(0.0359477 * W), int(0.96405229 * W)
        return 45, min(471, H), 41, min(601, W)
    return None


class ValidationStep:
    """``joint`` (default): image and mirror as one 2B-image forward when the model declares ``images_are_independent`` (GraphBins,
    AdaBins, a GraphedGraphBins captured for 2B images with ``object_group = B``); False: two forwards, as the reference issues
    them (A/B)."""

    def __init__(self, model, args, flip_tta: bool = True, joint: bool = True):
        self.model, self.args, self.flip_tta, self.joint = model, args, flip_tta, joint
        ds = args[args.basic.dataset]
        self.min_depth, self.max_depth = float(ds.min_depth), float(ds.max_depth)

    def _call(self, *a):
        """One forward.  A captured graph is called through ``checked``: its fp16 range guard is read here (this sequential step
        reads its result next anyway) and a tripped batch re-run on the bf16-pair capture; an eager model guards itself."""
        fn = getattr(self.model, "checked", None)
        return fn(*a) if fn is not None else self.model(*a)

    def _forward_pair(self, image: torch.Tensor):
        """(output of the un-mirrored forward, depth of the mirrored forward -- still mirrored, as the metric kernel wants it)."""
        B = image.shape[0]
        mirrored = image.flip(dims=[3])
        if not (self.joint and getattr(self.model, "images_are_independent", False) and _joint_fits(self.model, B)):
            first = self._call(image)
            if getattr(self.model, "static_image", None) is not None:
                # a captured graph hands out its STATIC result tensors (bin_edges): the mirror's replay would overwrite them
                first = type(first)(**{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in first._asdict().items()})
            return first, self._call(mirrored).depth_pred
        both = torch.cat([image, mirrored], dim=0)
        # the provider sees the mirrored images as images of their own, exactly as the reference's detector does (:173); the two
        # halves keep their own Nmax (object_group = B): bit for bit what two calls compute, up to batch-size-dependent kernel
        # dispatch (split-K, tile shapes)
        out = self.model(both, None, None, None, B) if _takes_group(self.model) else self._call(both)
        first = type(out)(**{k: (None if v is None else v[:B]) for k, v in out._asdict().items()})
        return first, out.depth_pred[B:]

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, depth_gt: torch.Tensor, first_image_id: int = 0):
        """-> (records [B, 10] fp32 on the device, model output namedtuple of the un-mirrored forward)."""
        if self.flip_tta:
            out, mirror = self._forward_pair(image)
        else:
            out, mirror = self._call(image), None
        H, W = depth_gt.shape[2:]
        rec = hip_ops.depth_metrics(out.depth_pred.contiguous(), depth_gt.contiguous(), self.min_depth, self.max_depth,
                                    crop=crop_box(self.args, H, W),
                                    pred_mirror=None if mirror is None else mirror.contiguous(),
                                    first_image_id=first_image_id)
        return rec, out


def _takes_group(model) -> bool:
    """Whether ``model.forward`` has the ``object_group`` argument.  A captured graph (``GraphedGraphBins``: ``__call__`` only,
    no ``forward``) does not -- its group was fixed when it was captured."""
    import inspect
    fwd = getattr(model, "forward", None)
    if fwd is None:
        return False
    try:
        return "object_group" in inspect.signature(fwd).parameters
    except (TypeError, ValueError):
        return False


def _joint_fits(model, B: int) -> bool:
    """A model of free shape takes any 2B-image batch.  A captured graph takes the joint [batch | mirrored batch] forward only
    if it was captured for exactly that: 2B images with ``object_group = B`` (each half keeps its own Nmax);
    a graph captured for B images serves the pair as two replays instead."""
    static = getattr(model, "static_image", None)
    if static is None:
        return True
    return int(static.shape[0]) == 2 * B and getattr(model, "object_group", None) == B


def hw_queue_note(slots: int) -> Optional[str]:
    """None when the environment gives ``slots`` concurrent streams hardware queues of their own, else what is wrong."""
    import os
    if slots <= 1:
        return None
    raw = os.environ.get("GPU_MAX_HW_QUEUES")
    try:
        have = int(raw) if raw is not None else None
    except ValueError:
        have = None
    if have is not None and have >= slots:
        return None
    return (f"{slots} steps in flight want GPU_MAX_HW_QUEUES >= {slots} set before the HIP runtime starts (found "
            f"{'unset: the runtime default of 4 shares queues between later streams' if raw is None else repr(raw)}); slots that "
            "share a hardware queue serialise (measured 781 vs 840 img/s)")


class PipelinedValidation:
    """The reference's validation loop -- one image at a time (main.py:58), model(image) and model(mirror) per image
    (modules/GraphBinsLM.py:159,173) -- with ``slots`` validation steps IN FLIGHT: each slot is a hipGraph of the joint
    [batch | mirrored batch] forward captured on a stream of its own (objcavit_amd/graph.py; live objects with
    ``object_capacity``), the steps go to the slots round-robin, the per-image records are collected at the end (or whenever the
    caller asks).  At bs 1 every launch is latency, so consecutive images overlap almost freely: measured on MI355X
    (bench.py --batch 1: the same slot mechanism) 288 img/s one after the other, 609 with three in flight, **685 - 692 with
    four** (the default; five and more collapse to 420 - 530 whatever GPU_MAX_HW_QUEUES says: the slots then share hardware queues);
    bs 2 (= image + mirror) 461 -> 817 -> 872.  Results are those of ``ValidationStep(joint=True)``: same kernels, same order per step.
    WANTS ``GPU_MAX_HW_QUEUES=4`` (= the default ``slots``) in the environment BEFORE the HIP runtime initialises: set explicitly the
    runtime deals every slot stream a hardware queue of its own (unset, its default of 4 gives only the first streams one: slots
    that share a queue run one after the other, measured 781 instead of 840 img/s); ``OCV_SET_HW_QUEUES=4`` before ``import
    objcavit_amd`` sets it, bench.py and tests/conftest.py do the same.  NOT more than 4: on 6+ queues a captured forward that forks
    side streams replays 3x slower (hip_ops.hw_queues_allow_forks), so the package then captures lone batches without forks,
    process-wide.  A smaller or unset value is accepted but WARNED about and recorded in ``hip_ops.ROUTE_REPORT["PipelinedValidation"]``.

        pv = PipelinedValidation(model, args, example_image)
        for i, (image, depth_gt) in enumerate(loader):      # bs 1, as the reference
            pv.submit(image.cuda(non_blocking=True), depth_gt.cuda(non_blocking=True), first_image_id=i)
        records = pv.collect()                                # [N, 10] per-image records, submission order
    """

    def __init__(self, model, args, example_image: torch.Tensor, slots: int = 4, object_capacity: Optional[int] = None,
                 flip_tta: bool = True):
        from .graph import GraphedGraphBins
        if slots < 1:
            raise ValueError("PipelinedValidation: slots must be >= 1")
        note = hw_queue_note(slots)
        if note:
            import warnings
            warnings.warn(f"PipelinedValidation: {note}", RuntimeWarning, stacklevel=2)
            hip_ops.ROUTE_REPORT["PipelinedValidation"] = note
        self.args, self.flip_tta = args, flip_tta
        ds = args[args.basic.dataset]
        self.min_depth, self.max_depth = float(ds.min_depth), float(ds.max_depth)
        self.B = int(example_image.shape[0])
        both = torch.cat([example_image, example_image.flip(dims=[3])], 0) if flip_tta else example_image
        # slot streams that do NOT share a hardware queue (checked: hip_ops.independent_streams; the runtime's own dealing put two of
        # four consecutive streams on one queue -- 349 instead of 441 validated img/s at bs 1, profiles/r06_stream_queues.txt)
        streams = hip_ops.independent_streams(slots, example_image.device) if slots > 1 else [None]
        self.graphs = [GraphedGraphBins(model, both, object_capacity=object_capacity, object_group=self.B if flip_tta else None,
                                        in_flight=slots, stream=streams[k]) for k in range(slots)]
        self._next = 0
        self._pending = []
        self.rerun_steps = 0                                 # steps re-run on bf16 pairs by collect() (fp16 range guard)

    @torch.no_grad()
    def submit(self, image: torch.Tensor, depth_gt: torch.Tensor, first_image_id: int = 0, object_features=None, object_xywh_list=None) -> None:
        """Enqueue one validation step (image [B, 3, H, W] as captured, ground truth [B, 1, H', W']) on the next slot's stream;
        returns at once.  ``object_features`` / ``object_xywh_list``: the objects of the 2B images [batch | mirrored batch] for a
        graph with ``object_capacity`` (default: the model's provider is asked, on the slot's stream)."""
        if tuple(image.shape[1:]) != tuple(self.graphs[0].static_image.shape[1:]) or image.shape[0] != self.B:
            raise ValueError(f"captured for images {(self.B,) + tuple(self.graphs[0].static_image.shape[1:])}, got {tuple(image.shape)}")
        g = self.graphs[self._next]
        self._next = (self._next + 1) % len(self.graphs)
        caller = torch.cuda.current_stream(image.device)
        g.stream.wait_stream(caller)                         # image / ground truth were produced on the caller's stream
        with torch.cuda.stream(g.stream):
            both = torch.cat([image, image.flip(dims=[3])], 0) if self.flip_tta else image
            out = g(both, object_features, object_xywh_list) if g.objects is not None else g(both)
            rec = self._records(out, depth_gt, first_image_id)
        for t in (image, depth_gt):
            t.record_stream(g.stream)                        # the caching allocator must not recycle them under the slot's launches
        # the step's inputs stay referenced until collect(): a step whose fp16 range guard tripped (``g.last_flag``, read there in ONE
        # host copy for all pending steps) is re-run from them on the bf16-pair capture
        self._pending.append((rec, g.stream, g.last_flag, g, (both, depth_gt, first_image_id, object_features, object_xywh_list)))

    def _records(self, out, depth_gt: torch.Tensor, first_image_id: int) -> torch.Tensor:
        H, W = depth_gt.shape[2:]
        B = self.B
        return hip_ops.depth_metrics(out.depth_pred[:B].contiguous(), depth_gt.contiguous(), self.min_depth, self.max_depth,
                                     crop=crop_box(self.args, H, W),
                                     pred_mirror=out.depth_pred[B:].contiguous() if self.flip_tta else None,
                                     first_image_id=first_image_id)

    def collect(self) -> torch.Tensor:
        """Wait for every submitted step; -> records [N * B, 10] in submission order (and forget them).  The steps' inputs are held
        until here (fp16 range guard: a tripped step is re-run on bf16 pairs): call it every few hundred steps on a long run."""
        if not self._pending:
            return torch.empty(0, 10)
        for p in self._pending:
            p[1].synchronize()
        recs = [p[0] for p in self._pending]
        flags = [p[2] for p in self._pending]
        if any(f is not None for f in flags):
            # fp16 range guard: one host read for all pending steps; a tripped step is re-run on its slot's bf16-pair capture
            # (captured once, on the first trip) from the inputs kept since submit()
            hit = torch.cat([f if f is not None else torch.zeros(1, dtype=torch.int32, device=recs[0].device) for f in flags]).cpu()
            for i in hit.nonzero().flatten().tolist():
                _, _, _, g, (both, depth_gt, first_id, of, ox) = self._pending[i]
                out = g.rerun_on_bf16(both, of, ox) if g.objects is not None else g.rerun_on_bf16(both)
                recs[i] = self._records(out, depth_gt, first_id)
                self.rerun_steps += 1
            torch.cuda.current_stream(recs[0].device).synchronize()
        out = torch.cat(recs, 0)
        self._pending = []
        return out


def totals(records: torch.Tensor) -> Dict[str, float]:
    """Pixel-total metrics over all images of a record table (the reference's non-running metric classes after an
    epoch): weights n_valid, the two RMSEs recombined through their squares."""
    r = records.double().cpu()
    n = r[:, 8]
    tot = float(n.sum()) if float(n.sum()) > 0 else 1.0
    out = {}
    for i, k in enumerate(RECORD_FIELDS[:8]):
        if k in ("rmse", "rmse_log"):
            out[k] = float(((r[:, i] ** 2) * n).sum() / tot) ** 0.5
        else:
            out[k] = float((r[:, i] * n).sum() / tot)
    out["n_valid"] = int(n.sum())
    return out
