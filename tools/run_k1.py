"""The 1x1 GEMMs of the low-resolution first convolutions (tap products, 9 Cout columns) and the skip-part convolutions
(bs = 16): split-bf16 implicit-GEMM convolution kernel (pre-split input) against the pointwise kernel of the encoder
(fp32 input, split on the fly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
cl = torch.channels_last


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, h, w, K, N) in [(16, 15, 20, 512, 9216), (16, 30, 40, 1024, 4608), (16, 60, 80, 512, 2304), (16, 120, 160, 256, 1152)]:
    x = torch.randn(B, K, h, w, device="cuda").contiguous(memory_format=cl)
    wt = torch.randn(N, K, 1, 1, device="cuda") * 0.02
    xs = hip_ops.split_act(x)
    hi, lo = hip_ops.prep_conv_weight(wt)
    pw = hip_ops.pointwise_weight(wt)
    a = timeit(lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, 1, 0, out_fp32=True))
    b = timeit(lambda: hip_ops.pointwise_nhwc(x, pw, None, 0))
    ya, yb = hip_ops.conv_nhwc_split(xs, hi, lo, None, 1, 0, out_fp32=True), hip_ops.pointwise_nhwc(x, pw, None, 0)
    M = B * h * w
    print(f"B{B} {h}x{w} {K}->{N}: conv kernel {a:.3f} ms ({6 * M * N * K / a / 1e9:.0f} TF/s issued, {M * N * 4 / a / 1e6:.0f} GB/s written) | "
          f"pointwise kernel {b:.3f} ms | diff {float((ya - yb).abs().max() / ya.abs().max()):.1e}")
