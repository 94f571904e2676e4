"""Sensitivity of the project-layer shape to K (channel camping?), gate and residual."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
cl = torch.channels_last
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
B, H, W = 16, 120, 160
for Ci, Co, gate, res in [(240,40,1,1),(240,40,0,0),(240,40,1,0),(240,40,0,1),(232,40,0,0),(248,40,0,0),(256,40,0,0),(272,40,0,0),(240,32,0,0),(240,64,0,0),(240,128,0,0)]:
    x = torch.randn(B,Ci,H,W,device="cuda").contiguous(memory_format=cl)
    w = torch.randn(Co,Ci,device="cuda")*0.05; b = torch.randn(Co,device="cuda")
    g = torch.rand(B,Ci,device="cuda") if gate else None
    r = torch.randn(B,Co,H,W,device="cuda").contiguous(memory_format=cl) if res else None
    sw = hip_ops.SplitWeight(w)
    dt = t(lambda: hip_ops.pointwise_nhwc(x,sw,b,0,gate=g,residual=r))
    M=B*H*W; byts=M*(Ci+Co*(2 if res else 1))*4
    dsum = t(lambda: x.sum())
    y = torch.empty_like(x)
    dcp = t(lambda: y.copy_(x))
    print(f"{Ci}->{Co} gate{gate} res{res}: {dt*1e6:7.1f} us {byts/dt/1e12:5.2f} TB/s | x.sum {dsum*1e6:7.1f} us {M*Ci*4/dsum/1e12:5.2f} TB/s | copy {dcp*1e6:7.1f} us {2*M*Ci*4/dcp/1e12:5.2f} TB/s")
