"""tap_interp_combine on the four decoder stages + a KITTI stage: time per launch and a SHA-1 of the output bytes, so that two
library builds (OCV_LIB_PATH) can be compared bit for bit from their logs."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
SH = [(16, 17, 22, 30, 40, 1024), (16, 30, 40, 60, 80, 512), (16, 60, 80, 120, 160, 256), (16, 120, 160, 240, 320, 128), (8, 88, 304, 176, 608, 128)]
cl = torch.channels_last
g = torch.Generator(device="cuda").manual_seed(5)
tot = 0.0
for (B, h, w, H, W, Co) in SH:
    z = torch.randn(B, 9 * Co, h, w, device="cuda", generator=g).contiguous(memory_format=cl)
    s = torch.randn(B, Co, H, W, device="cuda", generator=g).contiguous(memory_format=cl)
    b = torch.randn(Co, device="cuda", generator=g)
    fn = lambda: hip_ops.tap_interp_combine(z, s, b, (H, W), 2, out_fp32=True, out_split=True)
    for _ in range(3): y, ys = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot += ms
    hsh = hashlib.sha1(y.cpu().numpy().tobytes() + ys.hl.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"B{B} {h}x{w}->{H}x{W} Cout {Co}: {ms:.3f} ms  sha1 {hsh}")
print(f"sum {tot:.3f} ms  [{os.environ.get('OCV_LIB_PATH', 'product build')}]")
