"""(needs tools/diag/conv_direct.patch.txt applied: the direct form is not in the product -- profiles/r06_conv_direct.txt)
Round 6: the pre-split convolution / GEMM kernel's two routes on the decoder's launches, same process, alternating.
  route 0  conv_split_dma_kernel: accumulators parked in LDS, all eight wavefronts store, one tile per workgroup (rounds 2 - 5)
  route 1  the shipped rule (csrc/conv_igemm.hip launch_conv): conv_split_direct_kernel -- operands swapped, in-register epilogue, a
           workgroup walks npn = 2 or 3 channel tiles -- where that is modelled to pay, route 0 elsewhere; npnK = direct form forced
`python tools/ab_conv_direct.py [bs] [iters]` -> a table of us per launch (HIP events over `iters` back-to-back launches, three blocks per
route interleaved, best block) + whether the two routes' outputs are bit-identical, + an npn sweep on the multi-tile launches.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)


def timed(fn, n=IT):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def route(direct, npn=0):
    assert lib.ocv_conv_split_set_dispatch(int(direct), int(npn)) == 0


def ab(name, fn, out_of, sweeps=()):
    """fn() runs the launch(es); out_of(result) -> tensor to compare."""
    res = {}
    for key in [(0, 0), (1, 0)] + [(1, n) for n in sweeps]:
        route(*key)
        res[key] = [1e30, out_of(fn()).clone()]
    for _ in range(3):
        for key in res:
            route(*key)
            res[key][0] = min(res[key][0], timed(fn))
    route(1, 0)
    t0, y0 = res[(0, 0)]
    t1, y1 = res[(1, 0)]
    same = bool(torch.equal(y0, y1))
    dev_ = float((y0.float() - y1.float()).abs().max() / y0.float().abs().max().clamp_min(1e-30))
    extra = "  ".join(f"npn{n}: {res[(1, n)][0]:7.1f}" for n in sweeps)
    eq = all(torch.equal(res[(1, n)][1], y1) for n in sweeps)
    print(f"{name:44s} parked {t0:8.1f}  rule {t1:8.1f}  ({t1 / t0:5.3f}x)  bit-identical {same} (max dev {dev_:.1e})  {extra}{'' if eq else '  NPN MISMATCH'}", flush=True)
    return t0, t1


def split_input(Bn, C, H, W):
    x = torch.randn(Bn, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    return hip_ops.upsample_concat_split(x, None, (H, W), f16=True)


tot0 = tot1 = 0.0
print(f"# bs {B}, {IT} launches per block, fp16 pairs; us per launch")
# tap GEMMs of the low-resolution first convolutions: 1x1, K -> 9 Cout, raw fp32 out
for (h, w, K, Cout) in [(15, 20, 512, 1024), (30, 40, 1024, 512), (60, 80, 512, 256), (120, 160, 256, 128)]:
    xs = split_input(B, K, h, w)
    wt = torch.randn(9 * Cout, K, 1, 1, device=dev) * 0.02
    hi, lo, osc = hip_ops.prep_conv_weight(wt, f16=True)
    nt = 9 * Cout // 128
    sw = sorted({n for n in (1, 2, 3, 9, 18, nt) if n <= nt})
    a, b = ab(f"tap GEMM {h}x{w} {K}->{9 * Cout}", lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, 1, hip_ops.ACT_NONE, out_fp32=True, oscale=osc),
              lambda y: y, sweeps=sw)
    tot0 += a; tot1 += b
    del xs, wt, hi, lo
# skip-part convolutions: 3x3, fp32 out, no bias
for (H, W, Cin, Cout) in [(240, 320, 24, 128), (120, 160, 40, 256), (60, 80, 64, 512), (30, 40, 176, 1024)]:
    xs = split_input(B, Cin, H, W)
    hi, lo, osc = hip_ops.prep_conv_weight(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05, f16=True)
    nt = Cout // 128
    a, b = ab(f"skip conv {H}x{W} {Cin}->{Cout}", lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, 3, hip_ops.ACT_NONE, out_fp32=True, oscale=osc),
              lambda y: y, sweeps=[n for n in (1, 2, 4, 8) if n <= nt and nt > 1])
    tot0 += a; tot1 += b
    del xs, hi, lo
# second convolutions: 3x3 + bias + LeakyReLU, split output only
for (H, W, C) in [(120, 160, 256), (240, 320, 128)]:
    xs = split_input(B, C, H, W)
    hi, lo, osc = hip_ops.prep_conv_weight(torch.randn(C, C, 3, 3, device=dev) * 0.03, f16=True)
    bias = torch.randn(C, device=dev) * 0.1
    a, b = ab(f"conv3x3 {H}x{W} {C}->{C} leaky, pairs out", lambda: hip_ops.conv_nhwc_split(xs, hi, lo, bias, 3, hip_ops.ACT_LEAKY_RELU, out_fp32=False, out_split=True, oscale=osc),
              lambda y: y.hl, sweeps=[n for n in (1, 2) if C // 128 > 1])
    tot0 += a * (3 if C == 128 else 1); tot1 += b * (3 if C == 128 else 1)
    del xs, hi, lo
# Winograd F(4x4, 3x3): input transform + 36-GEMM batch + output transform (the batch is the kernel under test)
for (H, W, C) in [(30, 40, 1024), (60, 80, 512)]:
    xs = split_input(B, C, H, W)
    uh, ul, fs, cs = hip_ops.prep_winograd43_weight(torch.randn(C, C, 3, 3, device=dev) * 0.02)
    bias = torch.randn(C, device=dev) * 0.1
    a, b = ab(f"winograd43 {H}x{W} {C}->{C} (3 launches)", lambda: hip_ops.conv3x3_winograd43_split(xs, uh, ul, fs, bias, hip_ops.ACT_LEAKY_RELU, out_fp32=False, out_split=True, cscale=cs),
              lambda y: y.hl, sweeps=[n for n in (1, 2, 4, 8) if n <= C // 128])
    tot0 += a; tot1 += b
    del xs, uh, ul
print(f"# sum over one forward's launches of these shapes: parked {tot0:.0f} us, rule {tot1:.0f} us ({tot1 - tot0:+.0f})")
