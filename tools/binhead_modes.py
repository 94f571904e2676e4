"""The fused bin head (logits -> softmax -> depth, one pass over the NHWC map) under its arithmetic forms, HIP-event time of the
main launch(es): h2 (two-term fp16, scaled low term, all 256 bins per workgroup, two-level logits: the default) | h2dense (the same,
every 32-bin tile in full: round 4's kernel) | split3 (three-term bf16, two bin halves + merge) | exact (fp32 MFMA) -- on a peaked
softmax (logit std ~13: one or two tiles of eight kept) and on a flat one (every tile kept: the two-level form's worst case)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
shapes = [(16, 240, 320), (16, 480, 640), (8, 176, 608)]
torch.manual_seed(0)
for (B, h, w, gain) in [(B, h, w, g) for (B, h, w) in shapes for g in (0.2, 0.002)]:
    feat = torch.randn(B, 128, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    q = torch.randn(B, 128, 128, device="cuda") * 0.5
    wout, bout = torch.randn(256, 128, 1, 1, device="cuda") * gain, torch.randn(256, device="cuda") * 0.5
    centers = torch.rand(B, 256, device="cuda").cumsum(1)
    byts = B * (h * w * 128 * 4 + h * w * 4) + B * 256 * 128 * 4
    row, ref = [], None
    for mode in ("exact", "h2dense", "h2", "split3"):
        os.environ["OCV_BINHEAD"] = mode
        for _ in range(3):
            d = hip_ops.bin_head(feat, q, wout, bout, centers)
        torch.cuda.synchronize()
        hip_ops.enable_timing(True)
        for _ in range(10):
            d = hip_ops.bin_head(feat, q, wout, bout, centers)
        us = hip_ops.timing_results()["bin_head"][1] * 1e3
        hip_ops.enable_timing(False)
        if mode == "exact":
            ref = d
        row.append((mode, us, d))
    print(f"B={B} {h}x{w} gain {gain} ({byts / 1e6:.0f} MB): " + "  ".join(
        f"{m} {us:7.1f} us ({byts / us / 1e3:5.0f} GB/s, max rel vs exact {float(((d - ref).abs() / ref.abs()).max()):.1e})" for m, us, d in row))
