"""GraphBins with do_final_upscale (51 of the reference's 108 params files; features / tokens / depth at FULL resolution) at
480 x 640: ms per forward.  OCV_UPCONV=direct = the route before round 3 for the fifth stage (ATen resize + exact-fp32
convolution over the 131-channel concat)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
import gen
from objcavit_amd.config import make_args
from objcavit_amd.modules.GraphBins import GraphBins, SyntheticObjectProvider
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = make_args(language="clip", do_final_upscale=True)
m = GraphBins(args, object_provider=SyntheticObjectProvider(32, "clip", seed=42)).eval()
gen.load_into(m, 42, gen.PEAKY)
m = m.cuda()
img = gen.randn("img", (B, 3, 480, 640), 1).cuda()
for _ in range(3):
    out = m(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = m(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"do_final_upscale GraphBins bs={B} 480x640 [OCV_UPCONV={os.environ.get('OCV_UPCONV', 'lowres')}]: {dt * 1e3:.1f} ms per forward, "
      f"{B / dt:.1f} img/s, depth {tuple(out.depth_pred.shape)}")
