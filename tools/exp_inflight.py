"""Throughput with 1, 2 or 3 batches in flight: one hipGraph instance per slot (own static input, own scratch), replayed on
its own stream; slots take the steps round-robin.  Same work per step as bench.py (forward + fused validation metrics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from objcavit_amd import hip_ops
from objcavit_amd.graph import GraphedGraphBins
from objcavit_amd.validation import crop_box
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
model, sd, args = bench.build_model(dev)
B, H, W = 16, bench.H, bench.W
img = bench.synthetic_images(B, 42).to(dev)
gt = (torch.rand(B, 1, H, W) * 9.0 + 0.5).to(dev)
box = crop_box(args, H, W)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for nslot in (1, 2, 3):
    slots = [GraphedGraphBins(model, img) for _ in range(nslot)]
    streams = [torch.cuda.Stream() for _ in range(nslot)]
    def run(i):
        k = i % nslot
        with torch.cuda.stream(streams[k]):
            out = slots[k](slots[k].static_image)
            return hip_ops.depth_metrics(out.depth_pred, gt, 0.001, 10.0, crop=box, first_image_id=i * B)
    for i in range(2 * nslot): run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = [run(i) for i in range(steps)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ref = recs[0]
    same = all(torch.equal(r[:, :9], ref[:, :9]) for r in recs)
    print(f"{nslot} batch(es) in flight: {steps * B / dt:8.1f} img/s  {dt / steps * 1e3:7.3f} ms/step  (records identical across steps: {same})")
    del slots
