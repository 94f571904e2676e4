"""Per-entry-point GPU timings (event pairs) of the whole benchmark forward, aggregated by name."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from objcavit_amd import hip_ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model, _, _ = bench.build_model(dev)
img = bench.synthetic_images(16, 0).to(dev)
for _ in range(2): model(img)
torch.cuda.synchronize()
hip_ops.enable_timing(True)
n = 3
for _ in range(n): model(img)
res = hip_ops.timing_results()
hip_ops.enable_timing(False)
groups = {}
for k, (cnt, ms) in res.items():
    g = k.split("|")[0]
    groups.setdefault(g, [0, 0.0])
    groups[g][0] += cnt / n; groups[g][1] += cnt / n * ms
tot = sum(v[1] for v in groups.values())
print(f"timed total {tot:.2f} ms/step")
for g, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"{g:28s} n={c:5.0f} total={t:7.3f} ms")
if "-v" in sys.argv:
    for k, (cnt, ms) in sorted(res.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        print(f"  {k:44s} n={cnt // n:3d} each={ms * 1e3:8.1f} us")
