"""-m gpu: the device validation-step kernel (csrc/metrics.hip, row N2) through the C ABI against the CPU oracle
(oracle/validation_ref.py) and against the golden numbers produced by the reference's own metric classes."""
import pytest
import torch

import gen
from oracle import validation_ref as vr
from util import load_golden, rel_dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 2e-5          # fp32 per-pixel arithmetic on both sides; logf / log10f differ from the CPU's by ulps


@pytest.fixture(scope="module")
def ops():
    from objcavit_amd import hip_ops
    return hip_ops


def dev(t):
    return t.to("cuda")


@pytest.mark.parametrize("tag", list(gen.VALIDATION_CASES))
def test_depth_metrics_vs_reference_golden(ops, tag):
    from objcavit_amd.config import make_args
    from objcavit_amd.validation import crop_box, totals
    meta, z = load_golden(f"g6_validation_{tag}")
    gt, pa, pb = gen.validation_inputs(tag)
    args = make_args(dataset=meta["dataset"])
    args[meta["dataset"]].garg_crop, args[meta["dataset"]].eigen_crop = meta["garg"], meta["eigen"]
    box = crop_box(args, meta["H"], meta["W"])
    assert box == vr.crop_box(meta["dataset"], meta["garg"], meta["eigen"], meta["H"], meta["W"])
    rec = ops.depth_metrics(dev(pa), dev(gt), meta["min_depth"], meta["max_depth"], crop=box, pred_mirror=dev(pb), first_image_id=7)
    assert rec.shape == (meta["B"], 10)
    assert torch.equal(rec[:, 8].cpu(), torch.from_numpy(z["records"][:, 8]))         # valid-pixel counts: exact
    assert torch.equal(rec[:, 9].cpu(), torch.arange(7, 7 + meta["B"], dtype=torch.float32))
    assert rel_dev(rec[:, :8], z["records"][:, :8]) < TOL
    tot = totals(rec)
    for i, k in enumerate(vr.METRICS):
        assert abs(tot[k] - float(z["metrics"][i])) <= TOL * abs(float(z["metrics"][i])) + 1e-7, k
    assert torch.equal(rec, ops.depth_metrics(dev(pa), dev(gt), meta["min_depth"], meta["max_depth"], crop=box,
                                              pred_mirror=dev(pb), first_image_id=7))       # fixed-order reduction


@pytest.mark.parametrize("B,h,w,H,W,mirror", [(1, 1, 1, 1, 1, False), (2, 5, 7, 5, 7, True), (3, 11, 13, 37, 29, True),
                                              (16, 240, 320, 480, 640, False), (2, 30, 40, 31, 300, True)])
def test_depth_metrics_vs_oracle(ops, B, h, w, H, W, mirror):
    g = torch.Generator().manual_seed(B * 1000 + H)
    gt = torch.rand(B, 1, H, W, generator=g) * 12.0 - 1.0
    pa = torch.rand(B, 1, h, w, generator=g) * 11.0 + 0.1
    pb = torch.rand(B, 1, h, w, generator=g) * 11.0 + 0.1 if mirror else None
    if h > 2:
        pa[0, 0, 1, 1] = float("nan")
    if B > 1:
        gt[1] = -1.0                                   # an image without a single valid pixel
    ref = vr.per_image_records(pa, gt, 0.001, 10.0, depth_pred_mirror=pb, first_image_id=3)
    got = ops.depth_metrics(dev(pa), dev(gt), 0.001, 10.0, pred_mirror=None if pb is None else dev(pb), first_image_id=3)
    assert torch.equal(got[:, 8:].cpu(), ref[:, 8:])
    assert rel_dev(got[:, :8], ref[:, :8]) < TOL
    if B > 1:
        assert float(got[1, :8].abs().max()) == 0.0


def test_validation_step_flip_tta(ops):
    """ValidationStep on a stand-in model: two forwards (image, mirrored image), records equal the oracle's."""
    import collections
    from objcavit_amd.config import make_args
    from objcavit_amd.validation import ValidationStep
    Out = collections.namedtuple("Out", ["depth_pred", "bin_edges"])

    class Toy(torch.nn.Module):                        # depth depends on the horizontal position -> TTA is not a no-op
        def forward(self, image):
            d = image[:, :1, ::2, ::2].abs() * 3.0 + torch.linspace(0.5, 4.0, image.shape[3] // 2, device=image.device)
            return Out(d.contiguous(), None)

    args = make_args()
    img = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(1))
    gt = torch.rand(2, 1, 64, 96, generator=torch.Generator().manual_seed(2)) * 9.0 + 0.5
    m = Toy()
    rec, out = ValidationStep(m, args)(dev(img), dev(gt), first_image_id=10)
    ref = vr.per_image_records(m(img).depth_pred, gt, 0.001, 10.0, "nyu", False, True,
                               depth_pred_mirror=m(img.flip(dims=[3])).depth_pred, first_image_id=10)
    assert rel_dev(rec, ref) < TOL and out.depth_pred.shape == (2, 1, 32, 48)
