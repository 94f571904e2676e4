import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, gen
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
torch.set_grad_enabled(False)
m = GraphBins(make_args(strategy="learned", language="control_obj_zeros_512"), object_provider=SyntheticObjectProvider(16, "control_obj_zeros_512")).eval()
gen.load_into(m, 5, gen.PEAKY)
m = m.cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img = gen.randn("img", (8, 3, 480, 640), 5).cuda()[:B].contiguous()
def fin(t): return bool(torch.isfinite(t).all())
hooks = []
bad = []
def mk(name):
    def h(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        for o in outs:
            if isinstance(o, torch.Tensor) and not fin(o) and not bad:
                bad.append(name); print("first non-finite output:", name, tuple(o.shape), "nan count", int((~torch.isfinite(o)).sum()))
    return h
for name, mod in m.named_modules():
    if name: hooks.append(mod.register_forward_hook(mk(name)))
out = m(img)
print("depth finite:", fin(out.depth_pred), "bad:", bad[:1])
