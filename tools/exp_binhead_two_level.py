"""Where the two-level bin head (OCV_BINHEAD=h2) departs from the one-level kernel (h2dense): per logit gain, the share of pixels that
differ at all, the largest relative deviation, and for the worst pixel its top logits (fp64) with their 32-bin tiles -- to tell a
dropped tile that mattered from rounding along a different online-softmax path."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
from objcavit_amd.modules.AdaBins import bin_edges_and_centers
torch.manual_seed(3)
B, h, w = 2, 37, 53
g = lambda *s: torch.randn(*s, device="cuda")
for gain in (0.6, 6.0, 40.0):
    feat, q = g(B, 128, h, w).contiguous(memory_format=torch.channels_last), g(B, 128, 128) * 0.5
    wout, bout = g(256, 128, 1, 1) * gain / math.sqrt(128), g(256) * 0.5
    widths = torch.rand(B, 256, device="cuda") + 0.1
    _, centers = bin_edges_and_centers(widths / widths.sum(1, keepdim=True), 0.001, 10.0)
    out = {}
    for m in ("h2dense", "h2", "exact"):
        os.environ["OCV_BINHEAD"] = m
        out[m] = hip_ops.bin_head(feat, q, wout, bout, centers)
    Wf = wout.view(256, 128).double() @ q.double()                                       # [B, 256, 128]
    lg = torch.einsum("bkc,bcp->bkp", Wf, feat.double().reshape(B, 128, h * w)) + bout.double().view(1, 256, 1)
    d64 = (torch.softmax(lg, 1) * centers.double().unsqueeze(2)).sum(1).view(B, 1, h, w)
    rel = ((out["h2"] - out["h2dense"]).abs() / out["h2dense"])
    e = {m: float(((out[m].double() - d64).abs() / d64).max()) for m in out}
    print(f"gain {gain}: pixels that differ {100 * float((rel > 0).float().mean()):.1f} %, max rel dev {float(rel.max()):.2e}; "
          f"against fp64: dense {e['h2dense']:.2e}  two-level {e['h2']:.2e}  exact {e['exact']:.2e}")
    i = int(rel.flatten().argmax()); b, p = i // (h * w), i % (h * w)
    top = lg[b, :, p].topk(6)
    print("   worst pixel", (b, p // w, p % w), "depth dense / two-level / fp64:", float(out["h2dense"].flatten()[i]), float(out["h2"].flatten()[i]),
          float(d64.flatten()[i]), " top logits (nat) - max:", [round(float(v - top.values[0]), 2) for v in top.values], "tiles", [int(k) // 32 for k in top.indices],
          "centres", [round(float(centers[b, k]), 3) for k in top.indices])
