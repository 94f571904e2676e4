"""Do torch events recorded DURING hipGraph capture report elapsed times after replay (ROCm)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
x = torch.randn(4096, 4096, device="cuda")
y = torch.empty_like(x)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): torch.mm(x, x, out=y)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
try:
    with torch.cuda.graph(g):
        y.mul_(1.0)
        e0.record()
        torch.mm(x, x, out=y)
        e1.record()
        y.add_(1.0)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    print("elapsed in-graph events:", e0.elapsed_time(e1), "ms")
except Exception as ex:
    print("FAILED:", type(ex).__name__, ex)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.mm(x, x, out=y); b.record(); torch.cuda.synchronize(); print("eager mm:", a.elapsed_time(b), "ms")
