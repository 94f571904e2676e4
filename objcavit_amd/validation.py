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

    def _forward_pair(self, image: torch.Tensor):
        """(output of the un-mirrored forward, depth of the mirrored forward -- still mirrored, as the metric kernel wants it)."""
        B = image.shape[0]
        mirrored = image.flip(dims=[3])
        if not (self.joint and getattr(self.model, "images_are_independent", False)):
            return self.model(image), self.model(mirrored).depth_pred
        both = torch.cat([image, mirrored], dim=0)
        # the provider sees the mirrored images as images of their own, exactly as the reference's detector does (:173); the two
        # halves keep their own Nmax (object_group = B): bit for bit what two calls compute, up to batch-size-dependent kernel
        # dispatch (split-K, tile shapes)
        out = self.model(both, None, None, None, B) if _takes_group(self.model) else self.model(both)
        first = type(out)(**{k: (None if v is None else v[:B]) for k, v in out._asdict().items()})
        return first, out.depth_pred[B:]

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, depth_gt: torch.Tensor, first_image_id: int = 0):
        """-> (records [B, 10] fp32 on the device, model output namedtuple of the un-mirrored forward)."""
        if self.flip_tta:
            out, mirror = self._forward_pair(image)
        else:
            out, mirror = self.model(image), None
        H, W = depth_gt.shape[2:]
        rec = hip_ops.depth_metrics(out.depth_pred.contiguous(), depth_gt.contiguous(), self.min_depth, self.max_depth,
                                    crop=crop_box(self.args, H, W),
                                    pred_mirror=None if mirror is None else mirror.contiguous(),
                                    first_image_id=first_image_id)
        return rec, out


def _takes_group(model) -> bool:
    import inspect
    try:
        return "object_group" in inspect.signature(model.forward).parameters
    except (TypeError, ValueError):
        return False


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
