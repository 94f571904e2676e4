"""ocv_depth_metrics_fwd (partial + finish launches) at bs 16, 240x320 prediction against a 480x640 ground truth: us per call.
Round 5: 53 -> 37 us (wavefront xor-trees instead of one thread walking 256 LDS doubles per sum; the pixel loop unrolled by four)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from objcavit_amd import hip_ops
B,h,w,H,W=16,240,320,480,640
pred=torch.rand(B,1,h,w,device='cuda')*9+0.5
gt=torch.rand(B,1,H,W,device='cuda')*9+0.5
for _ in range(5): r=hip_ops.depth_metrics(pred,gt,0.001,10.0,crop=None,first_image_id=0)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): r=hip_ops.depth_metrics(pred,gt,0.001,10.0,crop=None,first_image_id=0)
e1.record(); torch.cuda.synchronize()
print("depth_metrics per call (two launches):", e0.elapsed_time(e1)/200*1e3, "us; record[0]:", r[0].tolist()[:4])
