"""Does the decoder's alpha-scaled intermediate grow with the input scale, and does a captured graph's guard word see it?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import gen
from objcavit_amd import hip_ops as ops
from objcavit_amd.graph import GraphedGraphBins
import test_hip_fp16_route as T
torch.set_grad_enabled(False)
H, W, B = 352, 384, 2
img = gen.randn("img", (B, 3, H, W), 41)
alpha = T._guard_alpha(ops, img, H, W)
m, sd, args = T._guard_model(H, W, alpha=alpha)
m(img.cuda())
print("alpha", alpha, "settled", m.dense_feature_extractor.decoder.settled_f16())
m.range_guard_sync = False
for s in (1.0, 2.0, 4.0, 8.0):
    ops.range_check(True)
    m((img * s).cuda())
    seen = {k: round(v[0], 1) for k, v in ops._Range.seen.items()}
    ops.range_check(False)
    print(f"x{s}: amax per fp16 tensor:", seen, flush=True)
m.range_guard_sync = True
for s in (1.0, 4.0):
    ops.ROUTE_REPORT.clear()
    m((img * s).cuda())
    print(f"eager guarded x{s}: report", dict(ops.ROUTE_REPORT), flush=True)
g = GraphedGraphBins(m, img.cuda())
for s in (1.0, 4.0, 8.0):
    g((img * s).cuda())
    print(f"graph x{s}: last_flag", int(g.last_flag.item()), "flag now", int(g.range_guard.flag.item()), flush=True)
