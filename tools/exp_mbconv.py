"""The fused expand + depthwise launch of the early MBConv blocks alone (ocv_mbconv_expand_dw_fwd + the squeeze-excite gate launch) at the bench's
three shapes: HIP-event time of the fused launch.  python tools/exp_mbconv.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
torch.manual_seed(0)
B = 16
for (Cin, mid, H, W, k, s) in ((40, 240, 120, 160, 3, 1), (24, 144, 240, 320, 3, 2), (64, 384, 60, 80, 3, 2)):
    x = torch.randn(B, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    we = hip_ops.SplitWeight(torch.randn(mid, Cin, device="cuda") * 0.2)
    be = torch.randn(mid, device="cuda") * 0.1
    wd = (torch.randn(k * k, mid, device="cuda") * 0.3).contiguous()
    bd = torch.randn(mid, device="cuda") * 0.1
    R = max(8, Cin // 4)
    se = (torch.randn(R, mid, device="cuda") * 0.1, torch.randn(R, device="cuda") * 0.1, torch.randn(R, mid, device="cuda") * 0.1, torch.randn(mid, device="cuda") * 0.1)
    f = lambda: hip_ops.expand_depthwise_se_gate(x, we, be, wd, bd, k, s, *se)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    hip_ops.enable_timing(True)
    for _ in range(30):
        f()
    t = hip_ops.timing_results()
    hip_ops.enable_timing(False)
    print(f"{Cin:3d} -> {mid:3d} at {H}x{W} k{k} s{s}: " + "  ".join(f"{n.split('|')[0]} {v[1] * 1e3:7.1f} us" for n, v in t.items()), flush=True)
