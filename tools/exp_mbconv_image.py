"""(Needs tools/diag/mbconv_image.hip.txt wired back in: see the end of that file.)  The late MBConv blocks' expand + depthwise (+ pooling) pair: ONE whole-image launch against the two-launch
form (pw_tile expand + dw_slide), per B5 shape at bs 16 and bs 1, HIP events, 30 launches per cell.  The squeeze-excite gate launch(es)
behind either form are timed separately (tiles = bands per image vs the depthwise kernel's partial rows)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops as ops, _lib
torch.set_grad_enabled(False)
SHAPES = [("stage4", 3, 30, 40, 128, 768), ("stage5", 5, 30, 40, 176, 1056), ("stage6", 5, 15, 20, 304, 1824), ("stage7", 3, 15, 20, 512, 3072)]
lib = _lib.load()


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


ONLY = sys.argv[1] if len(sys.argv) > 1 else None          # e.g. "stage6 16": one shape at one batch (for counter passes)
BS = (int(sys.argv[2]),) if len(sys.argv) > 2 else (16, 1)
for B in BS:
    print(f"bs {B}:   shape                 fused_us   expand_us  dw_us  two_launch_us   se(fused tiles)_us  se(dw tiles)_us")
    for name, k, H, W, Cin, mid in SHAPES:
        if ONLY and name != ONLY:
            continue
        R = Cin // 4
        x = (torch.randn(B, Cin, H, W, device="cuda")).contiguous(memory_format=torch.channels_last)
        we = ops.SplitWeight(torch.randn(mid, Cin, device="cuda") / math.sqrt(Cin))
        be = torch.randn(mid, device="cuda") * 0.3
        wd = (torch.randn(mid, 1, k, k, device="cuda") * 0.3).flatten(1).t().contiguous()
        bd = torch.randn(mid, device="cuda") * 0.2
        w1, b1 = torch.randn(R, mid, device="cuda") / math.sqrt(mid), torch.randn(R, device="cuda") * 0.3
        w2t, b2 = (torch.randn(mid, R, device="cuda") / math.sqrt(R)).t().contiguous(), torch.randn(mid, device="cuda") * 0.3
        out = torch.empty(B, mid, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        tiles_f = lib.ocv_mbconv_image_tiles(H, W, k)
        tiles_d = lib.ocv_depthwise_sum_tiles(B, mid, H, W, k, 1)
        part = torch.empty(B * max(tiles_f, tiles_d) * mid, device="cuda")
        gate, hid = torch.empty(B, mid, device="cuda"), torch.empty(B, R, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        fused = lambda: lib.ocv_mbconv_image_fwd(x.data_ptr(), we.packed.data_ptr(), be.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(),
                                                 part.data_ptr(), B, H, W, Cin, mid, k, st)
        y = ops.pointwise_nhwc(x, we, be, ops.ACT_SILU)
        expand = lambda: ops.pointwise_nhwc(x, we, be, ops.ACT_SILU)
        dw = lambda: lib.ocv_depthwise_conv_nhwc_sum_fwd(y.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), part.data_ptr(), B, mid, H, W,
                                                         k, 1, k // 2, k // 2, H, W, st)
        se = lambda t: lib.ocv_se_gate_partials_fwd(part.data_ptr(), t, H * W, w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(),
                                                    gate.data_ptr(), hid.data_ptr(), B, mid, R, st)
        tf, te, td = timeit(fused), timeit(expand), timeit(dw)
        print(f"  {name} k{k} {H}x{W} {Cin:4d}->{mid:4d}   {tf:8.1f}   {te:8.1f}  {td:6.1f}   {te + td:8.1f}        {timeit(lambda: se(tiles_f)):8.1f}   {timeit(lambda: se(tiles_d)):14.1f}", flush=True)
