"""The reference's validation step around the model (row N2 of SURVEY.md section 8f), on the device.

``ValidationStep(model, args)(image, depth_gt)`` does what ``GraphBinsLM.validation_step`` does between the batch and
its logged numbers (modules/GraphBinsLM.py:154-212): forward on the image and on its mirror, clamp, un-flip, average,
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
    def __init__(self, model, args, flip_tta: bool = True):
        self.model, self.args, self.flip_tta = model, args, flip_tta
        ds = args[args.basic.dataset]
        self.min_depth, self.max_depth = float(ds.min_depth), float(ds.max_depth)

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, depth_gt: torch.Tensor, first_image_id: int = 0):
        """-> (records [B, 10] fp32 on the device, model output namedtuple of the un-mirrored forward)."""
        out = self.model(image)
        mirror = self.model(image.flip(dims=[3])).depth_pred if self.flip_tta else None
        H, W = depth_gt.shape[2:]
        rec = hip_ops.depth_metrics(out.depth_pred.contiguous(), depth_gt.contiguous(), self.min_depth, self.max_depth,
                                    crop=crop_box(self.args, H, W),
                                    pred_mirror=None if mirror is None else mirror.contiguous(),
                                    first_image_id=first_image_id)
        return rec, out


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
