"""Kernel breakdown of encoder / decoder under a memory format, via torch.profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from objcavit_amd.config import make_args
from objcavit_amd.modules.DenseFeatureExtractor import DenseFeatureExtractor
torch.set_grad_enabled(False)
B = 16
cl = "cl" in sys.argv
part = "decoder" if "decoder" in sys.argv else "encoder"
m = DenseFeatureExtractor(make_args()).eval().cuda()
x = torch.randn(B, 3, 480, 640, device="cuda")
if cl:
    m = m.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
feats = m.encoder(x)
fn = (lambda: m.decoder(feats)) if part == "decoder" else (lambda: m.encoder(x))
for _ in range(2): fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): fn()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(f"{part} cl={cl}: total device time per iter {tot/3/1e3:.2f} ms")
for e in rows[:18]:
    print(f"{e.key[:90]:90s} n={e.count//3:4d} ms/iter={e.device_time_total/3/1e3:8.3f}")
