"""The skip-part convolutions alone, tap-major against packed taps (us per launch, HIP events over 20 launches, three interleaved blocks,
best block; max deviation between the two and from an fp64 convolution).  `python tools/run_skip_packed.py [bs]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from objcavit_amd import hip_ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (H, W, Cin, Cout) in [(240, 320, 24, 128), (120, 160, 40, 256), (60, 80, 64, 512), (30, 40, 176, 1024), (240, 320, 8, 128), (120, 160, 16, 64)]:
    x = torch.randn(B, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    xs = hip_ops.split_act(x, f16=True)
    hi, lo, osc = hip_ops.prep_conv_weight(w, f16=True)
    ph, pl, posc = hip_ops.prep_conv_weight_packed_taps(w, f16=True)
    dense = lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, 3, hip_ops.ACT_NONE, out_fp32=True, oscale=osc)
    packed = lambda: hip_ops.conv3x3_split_packed_taps(xs, ph, pl, None, hip_ops.ACT_NONE, out_fp32=True, oscale=posc)
    yd, yp = dense(), packed()
    td = tp = 1e30
    for _ in range(3):
        td = min(td, timed(dense)); tp = min(tp, timed(packed))
    ref = F.conv2d(x[:2].double(), w.double(), padding=1).float()
    m = float(ref.abs().max())
    print(f"B{B} {H}x{W} {Cin:3d}->{Cout:4d}: tap-major {td:7.1f} us  packed {tp:7.1f} us ({tp / td:5.3f}x; pays by rule: {hip_ops.packed_taps_pay(Cin)})  "
          f"|packed - tap-major| {float((yd - yp).abs().max()) / m:.1e}  |packed - fp64| {float((yp[:2] - ref).abs().max()) / m:.1e}", flush=True)
