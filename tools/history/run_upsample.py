"""Time the resize + concat + split kernel on the four decoder shapes (OCV_UPSAMPLE_QUAD=1: the 4-channel form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
B = 16
cl = torch.channels_last
tot = 0.0
for (h, w, C1, C2) in ((15, 20, 2048, 176), (30, 40, 1024, 64), (60, 80, 512, 40), (120, 160, 256, 24)):
    H, W = 2 * h, 2 * w
    x = torch.randn(B, C1, h, w, device="cuda").contiguous(memory_format=cl)
    s = torch.randn(B, C2, H, W, device="cuda").contiguous(memory_format=cl)
    for _ in range(3): y = hip_ops.upsample_concat_split(x, s, (H, W))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): y = hip_ops.upsample_concat_split(x, s, (H, W))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    Cp = (C1 + C2 + 31) // 32 * 32
    gb = (B * H * W * Cp * 4 + B * h * w * C1 * 4 + B * H * W * C2 * 4) / 1e9
    tot += ms
    print(f"{H}x{W} C{C1}+{C2}: {ms*1e3:.1f} us  {gb/ms:.2f} TB/s")
print(f"total {tot:.3f} ms")
