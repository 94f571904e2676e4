"""VERDICT r3 item 4, measure-before-build: the three launches of the F(4x4,3x3) convolution (input transform, 36-GEMM batch, output
transform) and the direct kernel on ONE long direct shape of the decoder, for a rocprofv3 kernel trace (one shape per process, so the
per-kernel averages of the trace belong to that shape):   python tools/exp_wino43_parts.py B H W Cin Cout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
B, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
torch.manual_seed(0)
x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
b = torch.randn(Co, device="cuda") * 0.1
xs = hip_ops.split_act(x, f16=True)
hi, lo, osc = hip_ops.prep_conv_weight(w, f16=True)
u = hip_ops.prep_winograd43_weight(w)
for _ in range(8):
    hip_ops.conv_nhwc_split(xs, hi, lo, b, 3, 2, out_fp32=False, out_split=True, oscale=osc)
    hip_ops.conv3x3_winograd43_split(xs, u[0], u[1], u[2], b, 2, out_fp32=False, out_split=True, cscale=u[3])
torch.cuda.synchronize()
print("done", B, H, W, Ci, Co)
