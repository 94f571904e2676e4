"""Run one split-bf16 conv shape a few times (for rocprofv3 --pmc)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
B, H, W, C1, C2, Cout = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (16, 60, 80, 1024, 64, 512))]
torch.manual_seed(0)
cl = torch.channels_last
x1 = torch.randn(B, C1, H, W, device="cuda").contiguous(memory_format=cl)
x2 = torch.randn(B, C2, H, W, device="cuda").contiguous(memory_format=cl) if C2 else None
w = torch.randn(Cout, C1 + C2, 3, 3, device="cuda") * 0.01
hi, lo = hip_ops.prep_conv_weight(w)
b = torch.zeros(Cout, device="cuda")
for _ in range(2): y = hip_ops.conv_nhwc(x1, x2, hi, lo, b, 3, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n): y = hip_ops.conv_nhwc(x1, x2, hi, lo, b, 3, 2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 2.0 * B * H * W * Cout * (C1 + C2) * 9
print(f"shape B{B} {H}x{W} C{C1}+{C2}->{Cout}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s fp32-equivalent ({3*fl/dt/1e12:.0f} TF/s bf16 issued)")
