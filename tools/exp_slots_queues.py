"""bs b forwards (+ metrics) with N graph slots in flight on streams from hip_ops.independent_streams, under GPU_MAX_HW_QUEUES = Q (set in
the environment before the run).  `GPU_MAX_HW_QUEUES=8 python tools/exp_slots_queues.py [batch] [slots ...]`"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from objcavit_amd import hip_ops
torch.set_grad_enabled(False)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
counts = [int(v) for v in sys.argv[2:]] or [4]
dev = torch.device("cuda:0")
model = bench.build_model(dev, bench.Workload(2, b))[0]
for n in counts:
    hip_ops.ROUTE_REPORT.pop("independent_streams", None)
    r = bench.side_leg(dev, bench.Workload(2, b), model, n, 1.0)
    print(f"GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']} bs {b}, {n} slots: {r['images_per_s']} img/s ({r['ms_per_step']} ms/step)"
          f"{'  [' + hip_ops.ROUTE_REPORT['independent_streams'] + ']' if 'independent_streams' in hip_ops.ROUTE_REPORT else ''}", flush=True)
