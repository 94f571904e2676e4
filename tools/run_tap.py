"""ocv_tap_interp_combine_fwd on the four decoder stages (bs = 16): time per launch, algorithmic HBM rate, and a check against
the definition (nine bilinear up-samplings of the tap products, shifted by their taps) on the smallest stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from objcavit_amd import hip_ops
SH = [(16, 17, 22, 30, 40, 1024), (16, 30, 40, 60, 80, 512), (16, 60, 80, 120, 160, 256), (16, 120, 160, 240, 320, 128)]
cl = torch.channels_last
for (B, h, w, H, W, Co) in SH:
    z = torch.randn(B, 9 * Co, h, w, device="cuda").contiguous(memory_format=cl)
    s = torch.randn(B, Co, H, W, device="cuda").contiguous(memory_format=cl)
    b = torch.randn(Co, device="cuda")
    fn = lambda: hip_ops.tap_interp_combine(z, s, b, (H, W), 2, out_fp32=False, out_split=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byts = z.numel() * 4 + 2 * s.numel() * 4
    line = f"B{B} {h}x{w}->{H}x{W} Cout {Co}: {ms:.3f} ms, {byts / ms / 1e6:.0f} GB/s algorithmic"
    if Co == 1024:
        y = hip_ops.tap_interp_combine(z, s, b, (H, W), 2)
        ref = s.double() + b.double().view(1, -1, 1, 1)
        for t in range(9):
            up = F.interpolate(z[:, t * Co:(t + 1) * Co].double(), size=(H, W), mode="bilinear", align_corners=True)
            ref = ref + F.pad(up, (1, 1, 1, 1))[:, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
        ref = F.leaky_relu(ref, 0.01)
        line += f" | max err / max {float((y.double() - ref).abs().max() / ref.abs().max()):.1e}"
    print(line)
