"""The decoder's deep 3x3 convolutions under their three forms at bs = 16: direct split-bf16, Winograd F(2x2,3x3) on two-term bf16
and Winograd F(4x4,3x3) on two-term fp16 (HIP events: all launches of the call), with the error of each against an fp64 convolution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from objcavit_amd import hip_ops
torch.manual_seed(0)

def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hip_ops.enable_timing(True)
    for _ in range(reps):
        fn()
    t = hip_ops.timing_results()
    hip_ops.enable_timing(False)
    return sum(v[1] for v in t.values()) * 1e3

for (B, H, W, Ci, Co) in ((16, 30, 40, 1024, 1024), (8, 22, 76, 1024, 1024), (16, 60, 80, 512, 512), (16, 120, 160, 256, 256)):
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    x = torch.where(x > 0, x, 0.01 * x)
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.01)
    xs = hip_ops.split_act(x)  # bf16 pairs (the F(2,2) form takes no others); the model feeds the F(4,3) form fp16 pairs: same kernel, other conversion
    hi, lo = hip_ops.prep_conv_weight(w)
    u2 = hip_ops.prep_winograd_weight(w)
    u4 = hip_ops.prep_winograd43_weight(w)
    forms = (("direct", lambda: hip_ops.conv_nhwc_split(xs, hi, lo, b, 3, 2, out_fp32=False, out_split=True)),
             ("F(2,2) bf16x2", lambda: hip_ops.conv3x3_winograd_split(xs, u2[0], u2[1], b, 2, out_fp32=False, out_split=True)),
             ("F(4,3) fp16x2", lambda: hip_ops.conv3x3_winograd43_split(xs, u4[0], u4[1], u4[2], b, 2, out_fp32=False, out_split=True, cscale=u4[3])))
    row = []
    for name, fn in forms:
        us = timeit(fn)
        y = fn().float()
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        row.append(f"{name} {us:7.1f} us (err {err:.1e})")
    print(f"B={B} {H}x{W} {Ci}->{Co}: " + "  ".join(row))
