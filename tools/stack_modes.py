"""A 4-layer nn.TransformerEncoder (E = 128, 4 heads, FF = 1024) through HipEncoderStack under OCV_TOKENS = h2 | split3 | fp32:
wall time per stack (torch events, 20 runs) at the token counts of the models."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from objcavit_amd.modules.layers import HipEncoderStack
torch.manual_seed(0)
enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(128, 4, 1024, batch_first=True), 4, enable_nested_tensor=False).eval().cuda()
for (B, S) in ((16, 300), (8, 418), (16, 1200), (16, 32)):
    x = torch.randn(B, S, 128, device="cuda")
    row, ref = [], None
    for mode in ("fp32", "split3", "h2"):
        os.environ["OCV_TOKENS"] = mode
        st = HipEncoderStack(enc)
        with torch.no_grad():
            for _ in range(3):
                y = st(x)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                y = st(x)
            t1.record()
            torch.cuda.synchronize()
        if ref is None:
            ref = y
        row.append(f"{mode} {t0.elapsed_time(t1) / 20 * 1e3:7.1f} us (max diff / max vs fp32 {float((y - ref).abs().max() / ref.abs().max()):.1e})")
    print(f"B={B} S={S}: " + "  ".join(row))
