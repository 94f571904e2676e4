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
steps = 30
island = (f"conv3x3|{B},{H // 2},{W // 2},280,128",)
for name, islands, copy, nslot in (("plain", (), False, 2), ("islands", island, False, 2), ("copy", (), True, 2), ("islands+copy", island, True, 2),
                                   ("islands+copy", island, True, 3), ("plain", (), False, 1)):
    slots = [GraphedGraphBins(model, img, eager_ops=islands) for _ in range(nslot)]
    streams = [g.stream for g in slots] if os.environ.get("OWN", "1") == "1" else [torch.cuda.Stream() for _ in range(nslot)]
    def run(i):
        k = i % nslot
        with torch.cuda.stream(streams[k]):
            out = slots[k](img if copy else slots[k].static_image)
            return hip_ops.depth_metrics(out.depth_pred, gt, 0.001, 10.0, crop=box, first_image_id=i * B)
    for i in range(2 * nslot): run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): run(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:14s} slots={nslot}: {steps * B / dt:8.1f} img/s  {dt / steps * 1e3:7.3f} ms/step   host submit {t_host / steps * 1e3:6.3f} ms/step")
    del slots, streams
