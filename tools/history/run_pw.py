"""Time pointwise conv shapes (B,H,W,Cin,Cout)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
shapes = [(16,240,320,24,144,3,0,0), (16,240,320,48,24,0,1,0), (16,120,160,40,240,3,0,0), (16,120,160,240,40,0,1,1), (16,60,80,384,64,0,1,1), (16,60,80,64,384,3,0,0), (16,30,40,768,128,0,1,1), (16,30,40,128,768,3,0,0), (16,15,20,3072,512,0,1,1), (16,15,20,512,3072,3,0,0), (16,15,20,1824,304,0,1,1), (16,15,20,304,1824,3,0,0), (16,30,40,1056,176,0,1,1), (16,30,40,176,1056,3,0,0), (16,240,320,24,24,0,1,1)]
cl = torch.channels_last
sel = [int(a) for a in sys.argv[1:]]
if sel: shapes = [shapes[i] for i in sel]
only_split = bool(os.environ.get("OCV_PW_CFG"))
for (B,H,W,Ci,Co,act,gate,res) in shapes:
    x = torch.randn(B,Ci,H,W,device="cuda").contiguous(memory_format=cl)
    w = torch.randn(Co,Ci,device="cuda")*0.05; b = torch.randn(Co,device="cuda")
    g = torch.rand(B,Ci,device="cuda") if gate else None
    r = torch.randn(B,Co,H,W,device="cuda").contiguous(memory_format=cl) if res else None
    M=B*H*W; byts=M*(Ci+Co*(2 if res else 1))*4; fl=2.0*M*Ci*Co
    out=[]
    for wt in ((hip_ops.SplitWeight(w),) if only_split else (w, hip_ops.SplitWeight(w))):
        for _ in range(2): y = hip_ops.pointwise_nhwc(x,wt,b,act,gate=g,residual=r)
        torch.cuda.synchronize(); n=10
        hip_ops.enable_timing(True)          # event pair around each launch: GPU time, not the Python call rate
        for _ in range(n): y = hip_ops.pointwise_nhwc(x,wt,b,act,gate=g,residual=r)
        dt = list(hip_ops.timing_results().values())[0][1] * 1e-3
        hip_ops.enable_timing(False)
        out.append(f"{dt*1e6:8.1f} us {byts/dt/1e12:5.2f} TB/s {fl/dt/1e12:6.1f} TF/s")
    print(f"M={M:8d} {Ci:5d}->{Co:5d} act{act} gate{gate} res{res}: " + (f"[{os.environ['OCV_PW_CFG']}] {out[0]}" if only_split else f"fp32 {out[0]} | split {out[1]}"))
