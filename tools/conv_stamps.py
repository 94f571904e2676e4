"""Diagnostic: per-phase cycle shares of conv_igemm (needs a build with OCV_EXTRA_HIPCC_FLAGS=-DOCV_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops, _lib
lib = _lib.load()
B, H, W, C1, C2, Cout = 16, 60, 80, 1024, 64, 512
cl = torch.channels_last
x1 = torch.randn(B, C1, H, W, device="cuda").contiguous(memory_format=cl)
x2 = torch.randn(B, C2, H, W, device="cuda").contiguous(memory_format=cl)
hi, lo = hip_ops.prep_conv_weight(torch.randn(Cout, C1 + C2, 3, 3, device="cuda") * 0.01)
for _ in range(3): hip_ops.conv_nhwc(x1, x2, hi, lo, None, 3, 2)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
fn = lib.ocv_conv_read_stamps; fn.restype = ctypes.c_int
print("rc", fn(out))
v = list(out)
n = max(v[7], 1)
print(f"steps {v[7]}; consumer: compute {v[0]/n:.0f} ticks/step, barrier wait {v[1]/n:.0f} ticks/step")
