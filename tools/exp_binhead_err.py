"""max-rel depth error of the e2e stress cases with the split vs the exact bin head (OCV_BINHEAD env, separate runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import gen
from oracle import restate
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins
torch.set_grad_enabled(False)
H, W = 352, 384
for seed in (77, 91, 5, 123):
    for kw, n_obj in ((dict(strategy="learned"), 16), (dict(strategy="grid_random"), 8)):
        args = make_args(dimensions_train=[H, W], dimensions_test=[H, W], language="clip", **kw)
        m = GraphBins(args).eval(); sd = gen.load_into(m, seed, gen.PEAKY)
        img = gen.randn("img", (2, 3, H, W), seed)
        feats = [gen.randn(f"f{i}", (n_obj, 512), seed, 10.0 / np.sqrt(512)) for i in range(2)]
        xywh = [gen.boxes(f"b{i}", n_obj, seed, H, W) for i in range(2)]
        out = m.cuda()(img.cuda(), [f.cuda() for f in feats], [b.cuda() for b in xywh])
        ref, _ = restate.graphbins_forward(img, feats, xywh, sd, 0.001, 10, **kw)
        e = ((out.depth_pred.cpu() - ref).abs() / ref)
        print(f"seed {seed} {kw['strategy']:12s} BINHEAD={os.environ.get('OCV_BINHEAD','split')}: max-rel {float(e.max()):.2e} mean-rel {float(e.mean()):.2e}")
