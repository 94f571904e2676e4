"""The encoder's 1x1 layers alone at bs 16 (ocv_pointwise_conv_nhwc_split_ws_fwd through hip_ops.pointwise_nhwc): HIP-event time per layer shape,
expand (SiLU) and project (gate + residual) forms, and the sum weighted by the layers' counts in EfficientNet-B5.  With OCV_LIB_PATH = a variant
library the same script times that build (A/B).  python tools/exp_pw.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
torch.manual_seed(0)
B = 16
# (H, W, Cin, Cout, kind, count per forward)
LAYERS = ((120, 160, 240, 40, "project", 5), (60, 80, 40, 240, "expand", 1), (60, 80, 240, 64, "project", 1), (60, 80, 64, 384, "expand", 4), (60, 80, 384, 64, "project", 4),
          (30, 40, 384, 128, "project", 1), (30, 40, 128, 768, "expand", 6), (30, 40, 768, 128, "project", 6), (30, 40, 768, 176, "project", 1),
          (30, 40, 176, 1056, "expand", 6), (15, 20, 1056, 304, "project", 1), (15, 20, 304, 1824, "expand", 8), (15, 20, 1824, 304, "project", 8),
          (15, 20, 1824, 512, "project", 1), (15, 20, 512, 3072, "expand", 2), (15, 20, 3072, 512, "project", 2))
total = 0.0
for (H, W, Ci, Co, kind, cnt) in LAYERS:
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = hip_ops.SplitWeight(torch.randn(Co, Ci, device="cuda") / Ci ** 0.5)
    b = torch.randn(Co, device="cuda") * 0.1
    if kind == "expand":
        f = lambda: hip_ops.pointwise_nhwc(x, w, b, hip_ops.ACT_SILU)
    else:
        gate = torch.rand(B, Ci, device="cuda")
        res = torch.randn(B, Co, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        f = lambda: hip_ops.pointwise_nhwc(x, w, b, hip_ops.ACT_NONE, gate=gate, residual=res)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    total += us * cnt
    print(f"{H:3d}x{W:<3d} {Ci:4d} -> {Co:4d} {kind:7s} x{cnt}: {us:7.1f} us", flush=True)
print(f"weighted sum: {total / 1e3:.3f} ms per forward")
