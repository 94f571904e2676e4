"""Cross-attention #1 (image <- objects, reference modules/ObjCAViT.py:195-201) against the HBM roofline at growing batch:
algorithmic bytes per SURVEY.md section 8d (3 S E 4 + S per image + the projection weights once per launch) / HIP-event
duration of ocv_mha_fwd, for S = 300 (NYU) and 418 (KITTI), 32 objects per image.  The north star asks for >= 40 % of
the HBM roofline for this kernel: the table says at which batch, if any, a launch gets there."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
E, H, N = 128, 4, 32
torch.manual_seed(0)
w = torch.randn(3 * E, E, device="cuda") * 0.1; b = torch.randn(3 * E, device="cuda") * 0.1
wo = torch.randn(E, E, device="cuda") * 0.1; bo = torch.randn(E, device="cuda") * 0.1
# round 3: `h2` = ocv_mha_few_keys_h2_fwd (every contraction a two-term fp16 split, K / V projected once per image: two launches,
# the time is their sum; round 5: `h2x1` = the same entry point's ONE-launch form, every query tile projecting K / V itself -- the
# default up to 512 workgroups), `split3` = ocv_mha_split3_fwd (three-term bf16 projections + exact-fp32 scores, same structure),
# `fp32` = round 2's single exact-fp32 launch (ocv_mha_fwd); same inputs, same process
from objcavit_amd import _lib
print(f"{'S':>4s} {'B':>5s} {'alg MB':>8s} {'MFLOP':>8s} | {'h2x1 us':>8s} | {'h2 us':>8s} {'GB/s':>7s} {'% 8TB/s':>8s} {'TFLOP/s':>8s} | {'split3 us':>9s} {'% 8TB/s':>8s} | {'fp32 us':>8s} {'% 8TB/s':>8s}")
for S in (300, 418):
    for B in (1, 2, 16, 32, 64, 128, 512, 2048):
        x = torch.randn(B, S, E, device="cuda")
        k = torch.full((B, S, E), 1e-4, device="cuda"); k[:, S - N:, :] = torch.randn(B, N, E, device="cuda")
        mask = torch.ones(B, S, dtype=torch.bool, device="cuda"); mask[:, :N] = False
        t = {}
        for tag in ("h2x1", "h2", "split3", "fp32"):
            if tag == "h2x1" and B > 512:
                t[tag] = float("nan")
                continue
            os.environ["OCV_TOKENS"] = "h2" if tag == "h2x1" else tag
            _lib.load().ocv_mha_few_keys_h2_set_dispatch((1 << 30) if tag == "h2x1" else 0)
            cache = {}
            for _ in range(3): hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=cache)
            torch.cuda.synchronize()
            hip_ops.enable_timing(True)
            for _ in range(20): hip_ops.mha(x, k, x, w, b, wo, bo, mask, H, kv_limit=N, packed=cache)
            t[tag] = hip_ops.timing_results()["mha_cross"][1] * 1e3
            hip_ops.enable_timing(False)
        byts = B * (3 * S * E * 4 + S) + 4 * E * E * 4 + 4 * E * 4
        flops = B * (4 * 2 * S * E * E + 2 * 2 * S * S * E)            # the reference's full-length form
        pct = lambda us: 100 * byts / us / 1e3 / 8000                  # noqa: E731
        print(f"{S:4d} {B:5d} {byts / 1e6:8.2f} {flops / 1e6:8.0f} | {t['h2x1']:8.1f} | {t['h2']:8.1f} {byts / t['h2'] / 1e3:7.0f} {pct(t['h2']):7.1f}% {flops / t['h2'] / 1e6:8.1f} | "
              f"{t['split3']:9.1f} {pct(t['split3']):7.1f}% | {t['fp32']:8.1f} {pct(t['fp32']):7.1f}%")
