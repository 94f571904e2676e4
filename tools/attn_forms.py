"""The self-attention core (softmax(q k^T / sqrt(d)) v, 4 heads of 32, packed QKV rows) on two-term fp16 splits against exact
fp32 MFMA: HIP-event time per launch at the token counts of the models (300 NYU, 418 KITTI, 1200 do_final_upscale) and 32 objects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
torch.manual_seed(0)
for (B, S) in ((16, 300), (8, 418), (16, 1200), (16, 32)):
    qkv = torch.randn(B, S, 384, device="cuda")
    q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
    res = {}
    for form in ("fp32", "h2"):
        os.environ["OCV_ATTN_FORM"] = form
        for _ in range(3):
            o = hip_ops.attention_core(q, k, v, None, 4)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            o = hip_ops.attention_core(q, k, v, None, 4)
        t1.record()
        torch.cuda.synchronize()
        res[form] = (t0.elapsed_time(t1) / 20 * 1e3, o)
    d = float((res["h2"][1] - res["fp32"][1]).abs().max() / res["fp32"][1].abs().max())
    print(f"B={B} S={S}: fp32 {res['fp32'][0]:7.1f} us  h2 {res['h2'][0]:7.1f} us  ({res['fp32'][0] / res['h2'][0]:.2f}x)  max diff / max {d:.1e}")
