"""The stem convolution alone (ocv_stem_conv_fwd: 3 -> 48, 3x3 stride 2, NCHW image -> NHWC, + bias + SiLU) at the bench's size: HIP-event time
and error against an fp64 convolution.  python tools/exp_stem.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from objcavit_amd import hip_ops
torch.manual_seed(0)
for B in (16, 1):
    x = torch.randn(B, 3, 480, 640, device="cuda")
    w = torch.randn(48, 3, 3, 3, device="cuda") * 0.2
    b = torch.randn(48, device="cuda") * 0.1
    for _ in range(5):
        y = hip_ops.stem_conv_same(x, w, b, 2, hip_ops.ACT_SILU)
    torch.cuda.synchronize()
    hip_ops.enable_timing(True)
    for _ in range(30):
        y = hip_ops.stem_conv_same(x, w, b, 2, hip_ops.ACT_SILU)
    t = hip_ops.timing_results()
    hip_ops.enable_timing(False)
    ref = F.silu(F.conv2d(F.pad(x[:1].double(), (0, 1, 0, 1)), w.double(), b.double(), stride=2))
    err = float((y[:1].double() - ref).abs().max() / ref.abs().max())
    mb = (x.numel() + y.numel()) * 4 / 1e6
    print(f"bs {B:2d}: {t['stem_conv'][1] * 1e3:7.1f} us  ({mb / t['stem_conv'][1] / 1e3:5.2f} TB/s of {mb:.0f} MB)  max err / max |y| vs fp64 {err:.1e}", flush=True)
