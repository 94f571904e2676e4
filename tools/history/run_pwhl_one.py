"""One late-stage 1x1 layer on the pre-split route, back to back (for rocprofv3 --pmc passes: tools/pmc_pwhl.sh).
Usage: python tools/run_pwhl_one.py M Cin Cout act [reps]   (OCV_PWHL_CFG / OCV_PWHL_PANEL select the kernel form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
M, Ci, Co, act = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
rpi = 1200 if M % 1200 == 0 else 300
B = M // rpi
x = torch.randn(B, Ci, rpi, 1, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(Co, Ci, device="cuda") * 0.05
b = torch.randn(Co, device="cuda")
xs = hip_ops.split_act(x)
sw = hip_ops.SplitWeight(w)
for _ in range(reps):
    y = hip_ops.pointwise_hl(xs, sw, b, act)
torch.cuda.synchronize()
hip_ops.enable_timing(True)
for _ in range(reps):
    y = hip_ops.pointwise_hl(xs, sw, b, act)
print(M, Ci, Co, {k: round(v[1] * 1e3, 1) for k, v in hip_ops.timing_results().items()}, "us")
