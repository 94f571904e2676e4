"""The fused bin head alone (ocv_bin_head_folded_ws_fwd, OCV_BINHEAD=h2): HIP-event time of the head launch at the bench's sizes and its
depth error against an fp64 evaluation of the reference's formula (GraphBins.py:109-119).  python tools/exp_binhead.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
if len(sys.argv) == 1:
    for v in ("r3", "r4", "r3", "r4"):
        print(f"OCV_BH_VARIANT={v}", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, OCV_BH_VARIANT=v), check=True)
    sys.exit(0)
import torch
from objcavit_amd import hip_ops
torch.manual_seed(0)
for B in (16, 8, 2, 1):
    h, w = 240, 320
    feat = torch.randn(B, 128, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    q = torch.randn(B, 128, 128, device="cuda") * 0.3
    wo = torch.randn(256, 128, 1, 1, device="cuda") * 0.3
    bo = torch.randn(256, device="cuda")
    cen = torch.sort(torch.rand(B, 256, device="cuda") * 10, dim=1).values
    for _ in range(5):
        d = hip_ops.bin_head(feat, q, wo, bo, cen)
    torch.cuda.synchronize()
    hip_ops.enable_timing(True)
    for _ in range(30):
        d = hip_ops.bin_head(feat, q, wo, bo, cen)
    t = hip_ops.timing_results()
    hip_ops.enable_timing(False)
    us = t["bin_head"][1] * 1e3
    n = min(B, 2)
    ram = torch.einsum("bchw,bqc->bqhw", feat[:n].double(), q[:n].double())
    logit = torch.einsum("bqhw,kq->bkhw", ram, wo.reshape(256, 128).double()) + bo.double()[None, :, None, None]
    ref = (torch.softmax(logit, 1) * cen[:n].double()[:, :, None, None]).sum(1, keepdim=True)
    err = float(((d[:n].double() - ref).abs() / ref.abs()).max())
    print(f"bs {B:2d}: {us:7.1f} us  ({B * h * w * 128 * 4 / us / 1e6:5.2f} TB/s of map)  max-rel vs fp64 {err:.2e}", flush=True)
