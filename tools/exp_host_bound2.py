import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # 1: run a bs-16 forward first (what bench.main has done before its legs)
wl = bench.Workload(2, b)
model = bench.build_model(dev, bench.Workload(2))[0]
if pre:
    img16 = bench.synthetic_images(16, 42, 480, 640).to(dev)
    for _ in range(2):
        model(img16)
    torch.cuda.synchronize()
for rep in range(2):
    r = bench.side_leg(dev, wl, model, 4, 1.2)
    print(f"pre={pre} side_leg bs {b} x4: {r['images_per_s']} img/s, {r['ms_per_step']} ms/step, steps {r['steps']}", flush=True)
r = bench.pipelined_validation_leg(dev, bench.Workload(2, 1), model, 4, 1.2)
print(f"pre={pre} PipelinedValidation bs1 x4: {r['images_per_s']} validated img/s, {r['ms_per_step']} ms/step", flush=True)
