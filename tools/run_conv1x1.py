"""Time the LDS-DMA split-bf16 convolution kernel (ksize = 1) on the late-stage 1x1 layer shapes: pre-split (hl32) input,
fp32 and/or split output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
SH = [(16, 30, 40, 128, 768), (16, 30, 40, 768, 128), (16, 30, 40, 176, 1056), (16, 30, 40, 1056, 176), (16, 15, 20, 304, 1824),
      (16, 15, 20, 1824, 304), (16, 15, 20, 512, 3072), (16, 15, 20, 3072, 512), (16, 15, 20, 512, 2048), (16, 15, 20, 2048, 2048)]
for (B, H, W, Ci, Co) in SH:
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    xs = hip_ops.upsample_concat_split(x, None, (H, W))
    hi, lo = hip_ops.prep_conv_weight(torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05)
    b = torch.zeros(Co, device="cuda")
    out = []
    for (of, os_) in ((True, False), (False, True), (True, True)):
        for _ in range(3):
            hip_ops.conv_nhwc_split(xs, hi, lo, b, 1, 3, out_fp32=of, out_split=os_)
        torch.cuda.synchronize()
        hip_ops.enable_timing(True)
        for _ in range(20):
            hip_ops.conv_nhwc_split(xs, hi, lo, b, 1, 3, out_fp32=of, out_split=os_)
        us = list(hip_ops.timing_results().values())[0][1] * 1e3
        hip_ops.enable_timing(False)
        out.append(us)
    print(f"M={B*H*W:6d} {Ci:5d}->{Co:5d}: fp32 out {out[0]:7.1f} us | split out {out[1]:7.1f} us | both {out[2]:7.1f} us")
