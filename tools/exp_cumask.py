"""Spatial partitioning experiment: HIP streams with CU masks (hipExtStreamCreateWithCUMask), every third CU each.
 (a) does the mask take effect (an MFMA-bound convolution should run ~3x longer on a third of the CUs)?
 (b) what does a memory-bound kernel lose on a third of the CUs?
 (c) convolution on one third + memory-bound kernel on another third, concurrently, against the two back to back on the
     whole chip."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(part, parts=3, ncu=256):
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(ncu):
        if cu % parts == part:
            mask[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value)


cl = torch.channels_last
torch.zeros(1, device="cuda")
B, H, W, C = 16, 120, 160, 256
xs = hip_ops.split_act(torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=cl))
hi, lo = hip_ops.prep_conv_weight(torch.randn(C, C, 3, 3, device="cuda") * 0.02)
conv = lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, 3, 2, out_fp32=False, out_split=True)
z = torch.randn(B, 9 * 128, 120, 160, device="cuda").contiguous(memory_format=cl)
s = torch.randn(B, 128, 240, 320, device="cuda").contiguous(memory_format=cl)
bias = torch.randn(128, device="cuda")
tap = lambda: hip_ops.tap_interp_combine(z, s, bias, (240, 320), 2, out_fp32=False, out_split=True)
streams = [masked_stream(i) for i in range(3)]
full = torch.cuda.Stream()


def run(fn, stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(2): fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): fn()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) / n


c_full, c_m = run(conv, full), run(conv, streams[0])
t_full, t_m = run(tap, full), run(tap, streams[1])
print(f"conv 256->256 @120x160: whole chip {c_full:.3f} ms, one third of the CUs {c_m:.3f} ms ({c_m / c_full:.2f}x)")
print(f"tap interpolation @240x320: whole chip {t_full:.3f} ms, one third of the CUs {t_m:.3f} ms ({t_m / t_full:.2f}x)")
# (c) concurrently: n convs on stream 0, and as many taps as fit in the same time on stream 1
import time
n = 10
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(streams[0]):
    for _ in range(n): conv()
with torch.cuda.stream(streams[1]):
    for _ in range(n): tap()
with torch.cuda.stream(streams[2]):
    for _ in range(n): tap()
torch.cuda.synchronize()
conc = (time.perf_counter() - t0) * 1e3
print(f"{n} convs (third 0) || {n} taps (third 1) || {n} taps (third 2): {conc:.2f} ms; back to back on the whole chip: "
      f"{n * (c_full + 2 * t_full):.2f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(full):
    for _ in range(n): conv(); tap(); tap()
torch.cuda.synchronize()
print(f"measured back to back on one whole-chip stream: {(time.perf_counter() - t0) * 1e3:.2f} ms")
