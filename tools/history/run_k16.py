"""The decoder's short-K launches (bs = 16) on the 32-channel-step DMA kernel (one workgroup per CU) and on the
16-channel-step variant (two per CU): ms per launch, bf16 matrix rate issued.  VERDICT r2 item 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
# (B, H, W, Cin, Cout, k)
SH = [(16, 240, 320, 24, 128, 3), (16, 120, 160, 40, 256, 3), (16, 60, 80, 64, 512, 3), (16, 30, 40, 176, 1024, 3),
      (16, 120, 160, 256, 1152, 1), (16, 60, 80, 512, 2304, 1), (16, 30, 40, 1024, 4608, 1), (16, 15, 20, 512, 9216, 1),
      (16, 240, 320, 128, 128, 3), (16, 120, 160, 256, 256, 3)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tot = {"0": 0.0, "1": 0.0}
print(f"{'shape':>34s} {'steps32':>7s} {'K32 ms':>8s} {'issued':>7s} | {'K16 ms':>8s} {'issued':>7s}   (issued = fraction of the 2.5 PF bf16 peak)")
for (B, H, W, Ci, Co, k) in SH:
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
    hi, lo = hip_ops.prep_conv_weight(w)
    xs = hip_ops.split_act(x)
    row = []
    for mode in ("0", "1"):
        os.environ["OCV_CONV_K16"] = mode
        for _ in range(3):
            hip_ops.conv_nhwc_split(xs, hi, lo, None, k, 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip_ops.conv_nhwc_split(xs, hi, lo, None, k, 0)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        Cp = (Ci + 31) // 32 * 32
        issued = 3 * 2.0 * B * H * W * Co * Cp * k * k / (ms * 1e-3) / 2.5e15
        row.append((ms, issued))
        if k * k * Cp // 32 <= 32:
            tot[mode] += ms
    print(f"B{B} {H}x{W} {Ci}->{Co} k{k}".rjust(34) + f" {k * k * ((Ci + 31) // 32):7d} {row[0][0]:8.3f} {row[0][1]:7.3f} | {row[1][0]:8.3f} {row[1][1]:7.3f}")
print(f"sum over the shapes with <= 32 steps: K32 {tot['0']:.3f} ms, K16 {tot['1']:.3f} ms")
