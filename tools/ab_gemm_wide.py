"""(needs tools/diag/gemm_wide.patch.txt applied: the wide kernel is not in the product -- profiles/r06_gemm_wide.txt)
Round 6: the tap GEMMs on the WIDE kernel (gemm_split_wide_kernel: 256 x 256 tile, eight multiplying wavefronts, two 64 KB buffers) against the
general 256 x 128 kernel, same process, interleaved blocks.  `python tools/ab_gemm_wide.py [bs]` -> us per launch, bit-identity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = _lib.load()
torch.manual_seed(0)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = [0.0, 0.0]
for (h, w, K, Cout) in [(15, 20, 512, 1024), (30, 40, 1024, 512), (60, 80, 512, 256), (120, 160, 256, 128)]:
    x = torch.randn(B, K, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    xs = hip_ops.split_act(x, f16=True)
    hi, lo, osc = hip_ops.prep_conv_weight(torch.randn(9 * Cout, K, 1, 1, device="cuda") * 0.02, f16=True)
    fn = lambda: hip_ops.conv_nhwc_split(xs, hi, lo, None, 1, hip_ops.ACT_NONE, out_fp32=True, oscale=osc)
    res = {}
    for on in (0, 1):
        lib.ocv_gemm_wide_set_dispatch(on)
        res[on] = [1e30, fn().clone()]
    for _ in range(3):
        for on in (0, 1):
            lib.ocv_gemm_wide_set_dispatch(on)
            res[on][0] = min(res[on][0], timed(fn))
    lib.ocv_gemm_wide_set_dispatch(1)
    tot[0] += res[0][0]; tot[1] += res[1][0]
    print(f"B{B} {h}x{w} {K}->{9 * Cout}: general {res[0][0]:7.1f} us  wide {res[1][0]:7.1f} us ({res[1][0] / res[0][0]:5.3f}x)  bit-identical {bool(torch.equal(res[0][1], res[1][1]))}", flush=True)
    del x, xs, hi, lo
print(f"# sum: general {tot[0]:.0f} us, wide {tot[1]:.0f} us")
