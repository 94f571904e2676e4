"""How far can a batch be scaled before the fp32 function itself is too ill-conditioned for a 1e-3 comparison?  For the range-guard
test (tests/test_hip_fp16_route.py GUARD_SCALE): deviation of the bf16-pair route and of the exact-fp32 route from the CPU oracle
at input scales 1 ... 1e4, and whether the fp16 pairs' guard trips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from objcavit_amd import hip_ops as ops, synth as gen
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
from oracle import restate
torch.set_grad_enabled(False)
H, W, B = 352, 384, 2
args = make_args(strategy="learned", language="clip", dimensions_train=[H, W], dimensions_test=[H, W])
m = GraphBins(args, object_provider=SyntheticObjectProvider(12, "clip", seed=3)).eval()
sd = gen.load_into(m, 41, gen.PEAKY)
m = m.cuda()
img = gen.randn("img", (B, 3, H, W), 41)
m(img.cuda())
dec = m.dense_feature_extractor.decoder
print("first batch:", dec.__dict__["_f16_modes"][dec._wkey()][2])
for s in (1.0, 30.0, 300.0, 3000.0, 1e4, 1e5):
    big = img * s
    feats, boxes, _ = m.object_provider(big.cuda())
    ref_d, _ = restate.graphbins_forward(big, [f.cpu() for f in feats], [b.cpu() for b in boxes], sd, 0.001, 10.0, strategy="learned")
    ops.ROUTE_REPORT.clear()
    d = m(big.cuda()).depth_pred.cpu()
    tripped = "range_guard" in ops.ROUTE_REPORT
    with ops.bf16_pairs():
        db = m(big.cuda()).depth_pred.cpu()
    rel = lambda a: float(((a - ref_d).abs() / ref_d).max())
    print(f"scale {s:8.0f}: guard tripped {tripped}; guarded route max-rel {rel(d):.2e}; bf16 pairs max-rel {rel(db):.2e}; depth range {float(ref_d.min()):.3f}..{float(ref_d.max()):.3f}", flush=True)
