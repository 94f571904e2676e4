"""Direct split-bf16 3x3 convolution against the Winograd F(2x2,3x3) form on the decoder shapes (bs = 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
SH = [(16, 30, 40, 2224, 1024), (16, 30, 40, 1024, 1024), (16, 60, 80, 1088, 512), (16, 60, 80, 512, 512), (16, 120, 160, 552, 256),
      (16, 120, 160, 256, 256)]
for (B, H, W, Ci, Co) in SH:
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    xs = hip_ops.upsample_concat_split(x, None, (H, W))
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.01
    b = torch.zeros(Co, device="cuda")
    hi, lo = hip_ops.prep_conv_weight(w)
    uh, ul = hip_ops.prep_winograd_weight(w)
    res = []
    for fn in (lambda: hip_ops.conv_nhwc_split(xs, hi, lo, b, 3, 2, out_fp32=False, out_split=True),
               lambda: hip_ops.conv3x3_winograd_split(xs, uh, ul, b, 2, out_fp32=False, out_split=True)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    d = hip_ops.conv_nhwc_split(xs, hi, lo, b, 3, 2); wv = hip_ops.conv3x3_winograd_split(xs, uh, ul, b, 2)
    err = float((d - wv).abs().max() / d.abs().max())
    print(f"B{B} {H}x{W} {Ci}->{Co}: direct {res[0]:.3f} ms | winograd {res[1]:.3f} ms ({100 * (1 - res[1] / res[0]):+.0f} % faster) | max diff / max {err:.1e}")
