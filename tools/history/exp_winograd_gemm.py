"""Winograd F(2x2,3x3) feasibility, measured before built: the 16 per-position GEMMs of the two deepest
decoder convolutions (30x40: 2224->1024 and 1024->1024 at bs=16 -> 4800 Winograd tiles) timed on the existing LDS-DMA
split-bf16 kernel at ksize=1 with 16x the rows (one launch standing for the batched launch), next to the direct 3x3 launch
and to plain copies of the transform traffic (V = 16 x tiles x Cin split-bf16 written + read, M = 16 x tiles x Cout fp32
written + read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for (Ci, Co) in ((2224, 1024), (1024, 1024)):
    B, H, W = 16, 30, 40
    x = torch.randn(B, Ci, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    xs = hip_ops.upsample_concat_split(x, None, (H, W))
    hi3, lo3 = hip_ops.prep_conv_weight(torch.randn(Co, Ci, 3, 3, device="cuda") * 0.01)
    b = torch.zeros(Co, device="cuda")
    t_direct = timed(lambda: hip_ops.conv_nhwc_split(xs, hi3, lo3, b, 3, 2, out_fp32=False, out_split=True))
    # 16 GEMMs of [4800 x Ci] x [Ci x Co]: one launch over 16 x 4800 rows
    v = torch.randn(16 * B, Ci, H // 2, W // 2, device="cuda").contiguous(memory_format=torch.channels_last)
    vs = hip_ops.upsample_concat_split(v, None, (H // 2, W // 2))
    hi1, lo1 = hip_ops.prep_conv_weight(torch.randn(Co, Ci, 1, 1, device="cuda") * 0.01)
    t_gemm = timed(lambda: hip_ops.conv_nhwc_split(vs, hi1, lo1, None, 1, 0, out_fp32=True, out_split=False))
    # transform traffic as plain copies (lower bounds for the transform kernels)
    Cp = (Ci + 31) // 32 * 32
    vbytes = 16 * 4800 * Cp * 4
    mbytes = 16 * 4800 * Co * 4
    src = torch.empty(vbytes // 4, dtype=torch.float32, device="cuda"); dst = torch.empty_like(src)
    t_v = timed(lambda: dst.copy_(src))            # read V-bytes + write V-bytes ~ (transform writes V) + (GEMM reads V: inside t_gemm)
    src2 = torch.empty(mbytes // 4, dtype=torch.float32, device="cuda"); dst2 = torch.empty_like(src2)
    t_m = timed(lambda: dst2.copy_(src2))
    # input transform kernel ~ read act (V/4) + write V  ~ 0.625 of a V copy; output transform ~ read M + write M/4 ~ 0.625 of an M copy
    est = t_gemm + 0.625 * t_v + 0.625 * t_m
    print(f"{Ci}->{Co} @30x40 bs16: direct 3x3 {t_direct:.3f} ms | 16 Winograd GEMMs {t_gemm:.3f} ms + input transform >= {0.625*t_v:.3f} ms "
          f"+ output transform >= {0.625*t_m:.3f} ms = {est:.3f} ms  ->  {100*(1-est/t_direct):.0f} % faster at best")
