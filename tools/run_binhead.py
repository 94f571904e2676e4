"""The fused bin head at bs = 16 (240x320 map, 256 bins): three-term split kernel (default) and exact-fp32 kernel, ms per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
from objcavit_amd.modules.AdaBins import bin_edges_and_centers
B, h, w = 16, 240, 320
feat = torch.randn(B, 128, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
q = torch.randn(B, 300, 128, device="cuda") * 0.5
wout, bout = torch.randn(256, 128, 1, 1, device="cuda") * 0.5, torch.randn(256, device="cuda") * 0.5
widths = torch.rand(B, 256, device="cuda") + 0.1
widths = widths / widths.sum(1, keepdim=True)
_, centers = bin_edges_and_centers(widths, 0.001, 10.0)
for name, kw in (("split3", {}), ("exact", {"exact": True})):
    fn = lambda: hip_ops.bin_head(feat, q[:, 1:129, :], wout, bout, centers, **kw)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"bin head {name}: {e0.elapsed_time(e1) / 20:.3f} ms")
a, b = hip_ops.bin_head(feat, q[:, 1:129, :], wout, bout, centers), hip_ops.bin_head(feat, q[:, 1:129, :], wout, bout, centers, exact=True)
print("split3 vs exact: max rel", float(((a - b).abs() / b.abs()).max()))
