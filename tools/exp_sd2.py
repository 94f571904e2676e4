import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, gen
from oracle import restate
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
torch.set_grad_enabled(False)
def mr(a, b): return float(((a.cpu().double() - b.cpu().double()).abs() / b.cpu().double().abs()).max())
m = GraphBins(make_args(strategy="learned", language="control_obj_zeros_512"), object_provider=SyntheticObjectProvider(16, "control_obj_zeros_512")).eval()
sd = gen.load_into(m, 5, gen.PEAKY)
m = m.cuda()
img = gen.randn("img", (8, 3, 480, 640), 5).cuda()
d = m(img).depth_pred
feats, boxes, _ = m.object_provider(img)
solo = m(img[3:4], [feats[3]], [boxes[3]]).depth_pred
sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
bad = [k for k in sd if not torch.equal(sd2[k].float(), sd[k].float())]
print("changed keys", bad[:5], len(bad))
ref1, _ = restate.graphbins_forward(img[3:4].cpu(), [feats[3].cpu()], [boxes[3].cpu()], sd, 0.001, 10, strategy="learned")
ref2, _ = restate.graphbins_forward(img[3:4].cpu(), [feats[3].cpu()], [boxes[3].cpu()], sd2, 0.001, 10, strategy="learned")
print("d-vs-ref1", mr(d[3:4], ref1), "d-vs-ref2", mr(d[3:4], ref2), "ref1-vs-ref2", mr(ref1, ref2), "solo-ref1", mr(solo, ref1))
