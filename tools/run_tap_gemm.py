"""One tap GEMM (the low-resolution first convolution's 1 x 1 with 9 Cout columns, fp16 pairs, raw fp32 out) a few times -- a driver for
tools/pmc_tap_gemm.sh.  `python tools/run_tap_gemm.py [B h w K Cout]` (default: the last decoder stage at bs 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
B, h, w, K, Cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 120, 160, 256, 128))]
torch.manual_seed(0)
x = torch.randn(B, K, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
xs = hip_ops.split_act(x, f16=True)
hi, lo, osc = hip_ops.prep_conv_weight(torch.randn(9 * Cout, K, 1, 1, device="cuda") * 0.02, f16=True)
for _ in range(int(os.environ.get("OCV_ITERS", "6"))):
    z = hip_ops.conv_nhwc_split(xs, hi, lo, None, 1, hip_ops.ACT_NONE, out_fp32=True, oscale=osc)
torch.cuda.synchronize()
print("ok", tuple(z.shape))
