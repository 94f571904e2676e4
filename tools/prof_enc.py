"""Per-shape launch timings of the NHWC encoder (pointwise / depthwise / everything else by torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
from objcavit_amd.config import make_args
from objcavit_amd.modules.DenseFeatureExtractor import DenseFeatureExtractor
torch.set_grad_enabled(False)
B = 16
m = DenseFeatureExtractor(make_args()).eval().cuda()
x = torch.randn(B, 3, 480, 640, device="cuda")
for _ in range(2): m.encoder(x)
hip_ops.enable_timing(True)
for _ in range(3): m.encoder(x)
res = hip_ops.timing_results()
hip_ops.enable_timing(False)
rows = sorted(((n / 3 * ms, n // 3, ms, k) for k, (n, ms) in res.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"timed total {tot:.2f} ms/iter")
for t, n, ms, k in rows[:40]:
    print(f"{k:40s} n={n:3d} each={ms*1e3:8.1f} us total={t:7.3f} ms")
