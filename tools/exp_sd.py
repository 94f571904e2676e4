import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, gen
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
torch.set_grad_enabled(False)
m = GraphBins(make_args(strategy="learned", language="control_obj_zeros_512"), object_provider=SyntheticObjectProvider(16, "control_obj_zeros_512")).eval()
sd = gen.load_into(m, 5, gen.PEAKY)
m = m.cuda()
img = gen.randn("img", (2, 3, 480, 640), 5).cuda()
m(img)
sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
bad = [(k, float((sd2[k].float() - sd[k].float()).abs().max())) for k in sd if not torch.equal(sd2[k].float(), sd[k].float())]
print("changed keys:", bad[:10], len(bad))
print(set(sd) ^ set(sd2))
