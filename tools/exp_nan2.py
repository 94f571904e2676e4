import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, gen
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
torch.set_grad_enabled(False)
def fin(t): return bool(torch.isfinite(t).all())
def first(kw, n_obj):
    H, W = 352, 384
    args = make_args(dimensions_train=[H, W], dimensions_test=[H, W], language="clip", **kw)
    m = GraphBins(args).eval(); gen.load_into(m, 77, gen.PEAKY); m = m.cuda()
    img = gen.randn("img", (2, 3, H, W), 77)
    feats = [gen.randn(f"f{i}", (n_obj, 512), 77, 10.0 / np.sqrt(512)).cuda() for i in range(2)]
    xywh = [gen.boxes(f"b{i}", n_obj, 77, H, W).cuda() for i in range(2)]
    out = m(img.cuda(), feats, xywh); print("mini", kw, fin(out.depth_pred))
which = sys.argv[1:] or ["learned"]
if "learned" in which: first(dict(strategy="learned"), 16)
if "2saca" in which: first(dict(strategy="learned_bbox_wh", use_2_saca=True), 90)
if "grid" in which: first(dict(strategy="grid_random"), 8)
if "roi" in which: first(dict(strategy="grid_random_roi_align"), 5)
gc.collect(); torch.cuda.empty_cache() if "empty" in which else None
m = GraphBins(make_args(strategy="learned", language="control_obj_zeros_512"), object_provider=SyntheticObjectProvider(16, "control_obj_zeros_512")).eval()
gen.load_into(m, 5, gen.PEAKY); m = m.cuda()
img = gen.randn("img", (8, 3, 480, 640), 5).cuda()
bad = []
def mk(name):
    def h(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        for o in outs:
            if isinstance(o, torch.Tensor) and not fin(o) and not bad:
                ins = [fin(i) for i in inp if isinstance(i, torch.Tensor)]
                bad.append(name); print("first non-finite:", name, type(mod).__name__, tuple(o.shape), "bad elems", int((~torch.isfinite(o)).sum()), "inputs finite", ins)
    return h
for name, mod in m.named_modules():
    if name: mod.register_forward_hook(mk(name))
out = m(img)
print("depth finite:", fin(out.depth_pred))
