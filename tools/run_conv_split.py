"""Run one pre-split conv shape a few times (for rocprofv3 --pmc / stamps)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops, _lib
B, H, W, Cin, Cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 60, 80, 1088, 512))]
F16 = os.environ.get("OCV_RUN_F16", "1") == "1"            # fp16 pairs (the product's default) or bf16 pairs
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
if os.environ.get("OCV_RUN_ZEROS") == "1":                 # all-zero activations: the same instruction stream with operands that do not toggle
    x.zero_()
xs = hip_ops.upsample_concat_split(x, None, (H, W), f16=F16)
prep = hip_ops.prep_conv_weight(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.01, f16=F16)
hi, lo, osc = (prep + (None,))[:3]
b = torch.zeros(Cout, device="cuda")
for _ in range(2): y = hip_ops.conv_nhwc_split(xs, hi, lo, b, 3, 2, out_fp32=False, out_split=True, oscale=osc)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = int(os.environ.get('OCV_ITERS', '5'))
for _ in range(n): y = hip_ops.conv_nhwc_split(xs, hi, lo, b, 3, 2, out_fp32=False, out_split=True, oscale=osc)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 2.0 * B * H * W * Cout * Cin * 9
print(f"shape B{B} {H}x{W} C{Cin}->{Cout} split-in: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s fp32-equivalent ({3*fl/dt/1e12:.0f} TF/s bf16 issued)")
lib = _lib.load()
if hasattr(lib, "ocv_conv_read_stamps"):
    out = (ctypes.c_ulonglong * 16)(); lib.ocv_conv_read_stamps.restype = ctypes.c_int; lib.ocv_conv_read_stamps(out)
    v = list(out); nn = max(v[7], 1)
    print(f"stamps: steps {v[7]}; consumer compute {v[0]/nn:.0f} ticks/step, barrier wait {v[1]/nn:.0f} ticks/step")
    print(f"tile of workgroup 0: prologue {v[8]} ticks, epilogue {v[9]}, whole {v[10]}  (loop {v[0]+v[1]})")
    if v[13]:
        print(f"K loop of that workgroup: {v[12]} shader cycles in {v[13] * 10} ns of wall clock = {v[12] / (v[13] * 10.0):.2f} GHz "
              f"({v[12] / nn:.0f} cycles = {v[13] * 10.0 / nn:.0f} ns per step); K loop + epilogue {v[14] * 10} ns")
    h = nn / 2
    print(f"producer group 0 (per own interval): convert-interval: wait+convert {v[2]/h:.0f}, barrier {v[3]/h:.0f} | write-interval: lds write {v[4]/h:.0f}, issue {v[5]/h:.0f}, barrier {v[6]/h:.0f}")
