"""Poison the caching allocator with NaN, then run convs and compare with MIOpen (finds uninitialised reads / unwritten outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from objcavit_amd import hip_ops
torch.manual_seed(0)
cl = torch.channels_last
def poison(gb=24):
    t = torch.full((gb * 256 * 1024 * 1024,), float("nan"), device="cuda")
    del t
def check(B, H, W, C1, C2, Cout, k=3):
    x1 = torch.randn(B, C1, H, W, device="cuda").contiguous(memory_format=cl)
    x2 = torch.randn(B, C2, H, W, device="cuda").contiguous(memory_format=cl) if C2 else None
    w = torch.randn(Cout, C1 + C2, k, k, device="cuda") * 0.02
    b = torch.randn(Cout, device="cuda")
    hi, lo = hip_ops.prep_conv_weight(w)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.leaky_relu(F.conv2d(xin, w, b, padding=k // 2), 0.01)
    poison()
    for it in range(3):
        y = hip_ops.conv_nhwc(x1, x2, hi, lo, b, k, 2)
        bad = int((~torch.isfinite(y)).sum())
        err = float((y - ref).abs().max() / ref.abs().max()) if bad == 0 else float("nan")
        print(f"B{B} {H}x{W} {C1}+{C2}->{Cout} k{k} iter {it}: nonfinite {bad}, rel err {err:.2e}", flush=True)
check(8, 120, 160, 512, 40, 256)
check(8, 120, 160, 256, 0, 256)
check(8, 240, 320, 256, 24, 128)
check(8, 30, 40, 2048, 176, 1024)
