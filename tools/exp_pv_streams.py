"""Does PipelinedValidation's rate depend on WHICH pool streams its slots get?  `python tools/exp_pv_streams.py K`: K torch streams are
taken from the pool (and kept) before the four slots are built; prints validated img/s at bs 1 (image + mirror per step, four slots).
Fresh process per K (the pool's state is per process)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from objcavit_amd.validation import PipelinedValidation
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
wl = bench.Workload(2, 1)
model = bench.build_model(dev, wl)[0]
img = bench.synthetic_images(1, 42, wl.H, wl.W).to(dev)
gt = (torch.rand(1, 1, wl.H, wl.W) * 9 + 0.5).to(dev)
with torch.no_grad():
    model(torch.cat([img, img.flip(dims=[3])], 0))
held = [torch.cuda.Stream() for _ in range(K)]
pv = PipelinedValidation(model, model.args, img, slots=4)
for k in range(8):
    pv.submit(img, gt, first_image_id=k)
pv.collect()
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.6:
        for k in range(40):
            pv.submit(img, gt, first_image_id=n + k)
        pv.collect()
        n += 40
    best = max(best, n / (time.perf_counter() - t0))
from objcavit_amd import hip_ops
print(hip_ops.ROUTE_REPORT.get("independent_streams"))
print(f"K={K}: {best:.1f} validated img/s; slot streams {[int(g.stream.cuda_stream) % 100000 for g in pv.graphs]}", flush=True)
