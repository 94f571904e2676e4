"""solo vs in-batch vs CPU oracle for the config-2 property test (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, gen
from oracle import restate
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
torch.set_grad_enabled(False)
args = make_args(strategy="learned", language="control_obj_zeros_512")
m = GraphBins(args, object_provider=SyntheticObjectProvider(16, "control_obj_zeros_512")).eval()
sd = gen.load_into(m, 5, gen.PEAKY)
m = m.cuda()
img = gen.randn("img", (8, 3, 480, 640), 5)
g = img.cuda()
feats, boxes, _ = m.object_provider(g)
ref, _ = restate.graphbins_forward(img[3:4], [feats[3].cpu()], [boxes[3].cpu()], sd, 0.001, 10, strategy="learned")
def mr(a, b): return float(((a.cpu().double() - b.cpu().double()).abs() / b.cpu().double().abs()).max())
for mode in ("split_bf16", "miopen"):
    os.environ["OCV_CONV"] = mode
    d = m(g).depth_pred
    solo = m(g[3:4], [feats[3]], [boxes[3]]).depth_pred
    print(f"{mode:10s} batch-vs-cpu {mr(d[3:4], ref):.2e}  solo-vs-cpu {mr(solo, ref):.2e}  solo-vs-batch {mr(solo, d[3:4]):.2e}", flush=True)
