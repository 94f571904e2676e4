"""Cross-attention #1 (image <- objects, reference modules/ObjCAViT.py:195-201) against the HBM roofline at growing batch:
algorithmic bytes per SURVEY.md section 8d (3 S E 4 + S per image + the projection weights once per launch) / HIP-event
duration of ocv_mha_fwd, for S = 300 (NYU) and 418 (KITTI), 32 objects per image.  The north star asks for >= 40 % of
the HBM roofline for this kernel: the table says at which batch, if any, a launch gets there."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ["OCV_XATTN_SPLIT3_MIN_TILES"] = "0"      # measure the split3 form at every batch
from objcavit_amd import hip_ops
E, H, N = 128, 4, 32
torch.manual_seed(0)
w = torch.randn(3 * E, E, device="cuda") * 0.1; b = torch.randn(3 * E, device="cuda") * 0.1
wo = torch.randn(E, E, device="cuda") * 0.1; bo = torch.randn(E, device="cuda") * 0.1
# round 3: `split3` = ocv_mha_split3_fwd (K / V projected once per image + packed three-term-split projections, two launches:
# the time is their sum), `fp32` = round 2's single exact-fp32 launch (ocv_mha_fwd), same inputs
print(f"{'S':>4s} {'B':>5s} {'split3 us':>10s} {'alg MB':>8s} {'GB/s':>8s} {'% of 8 TB/s':>12s} {'MFLOP':>8s} {'TFLOP/s':>8s} | {'fp32 us':>8s} {'% of 8 TB/s':>12s}")
cache = {}
for S in (300, 418):
    for B in (16, 64, 128, 512, 2048):
        x = torch.randn(B, S, E, device="cuda")
        k = torch.full((B, S, E), 1e-4, device="cuda"); k[:, S - N:, :] = torch.randn(B, N, E, device="cuda")
        mask = torch.ones(B, S, dtype=torch.bool, device="cuda"); mask[:, :N] = False
        t = {}
        for tag, pk in (("split3", cache), ("fp32", None)):
            for _ in range(3): hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=pk)
            torch.cuda.synchronize()
            hip_ops.enable_timing(True)
            for _ in range(20): hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=pk)
            t[tag] = hip_ops.timing_results()["mha_cross"][1] * 1e3
            hip_ops.enable_timing(False)
        us = t["split3"]
        byts = B * (3 * S * E * 4 + S) + 4 * E * E * 4 + 4 * E * 4
        flops = B * (4 * 2 * S * E * E + 2 * 2 * S * S * E)            # the reference's full-length form
        print(f"{S:4d} {B:5d} {us:10.1f} {byts / 1e6:8.2f} {byts / us / 1e3:8.1f} {100 * byts / us / 1e3 / 8000:11.1f}% {flops / 1e6:8.0f} {flops / us / 1e6:8.2f} | "
              f"{t['fp32']:8.1f} {100 * byts / t['fp32'] / 1e3 / 8000:11.1f}%")
