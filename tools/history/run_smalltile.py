"""conv_split_dma_kernel tile height: every 1x1 / 3x3 split-bf16 launch of the decoder at bs = 16, timed in THIS process'
setting of OCV_CONV_SMALL_TILE (0 = 256-row tile, one workgroup per CU; 1 = 128-row tile, two per CU).  Run twice:
  OCV_CONV_SMALL_TILE=0 python tools/run_smalltile.py ; OCV_CONV_SMALL_TILE=1 python tools/run_smalltile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
cl = torch.channels_last
SH = [  # B, H, W, Cin, Cout, k, what
    (16, 15, 20, 512, 9216, 1, "stage-1 tap GEMM"), (16, 30, 40, 1024, 4608, 1, "stage-2 tap GEMM"),
    (16, 60, 80, 512, 2304, 1, "stage-3 tap GEMM"), (16, 120, 160, 256, 1152, 1, "stage-4 tap GEMM"),
    (16, 30, 40, 176, 1024, 3, "stage-1 skip part"), (16, 60, 80, 64, 512, 3, "stage-2 skip part"),
    (16, 120, 160, 40, 256, 3, "stage-3 skip part"), (16, 240, 320, 24, 128, 3, "stage-4 skip part"),
    (16, 240, 320, 128, 128, 3, "128->128"), (16, 120, 160, 256, 256, 3, "256->256"), (16, 60, 80, 512, 512, 3, "512->512")]
print("OCV_CONV_SMALL_TILE =", os.environ.get("OCV_CONV_SMALL_TILE", "(auto)"))
for (B, H, W, Ci, Co, k, what) in SH:
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=cl)
    xs = hip_ops.split_act(x)
    w = torch.randn(Co, Ci, k, k, device="cuda") * 0.02
    hi, lo = hip_ops.prep_conv_weight(w)
    split_out = what[0].isdigit()
    fn = lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, k, 0, out_fp32=not split_out, out_split=split_out)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    M = B * H * W
    print(f"{what:18s} B{B} {H}x{W} {Ci}->{Co} k{k}: {ms:.3f} ms  {6 * M * Co * Ci * k * k / ms / 1e9:.0f} TF/s issued")
