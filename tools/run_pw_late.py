"""Late-stage 1x1 layers (stages 4-7, conv_head, decoder conv2) at bs = 16: the 32-row tile kernel against each big-tile
shape, per layer.  Usage: python tools/run_pw_late.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops, _lib
M4, M6 = 19200, 4800
# (M, Cin, Cout, act, gate, res, launches per step)
LAYERS = [(M4, 384, 128, 0, 1, 0, 1), (M4, 128, 768, 3, 0, 0, 7), (M4, 768, 128, 0, 1, 1, 6), (M4, 768, 176, 0, 1, 0, 1),
          (M4, 176, 1056, 3, 0, 0, 7), (M4, 1056, 176, 0, 1, 1, 6), (M6, 1056, 304, 0, 1, 0, 1), (M6, 304, 1824, 3, 0, 0, 9),
          (M6, 1824, 304, 0, 1, 1, 8), (M6, 1824, 512, 0, 1, 0, 1), (M6, 512, 3072, 3, 0, 0, 2), (M6, 3072, 512, 0, 1, 1, 2),
          (M6, 512, 2048, 0, 0, 0, 1), (M6, 2048, 2048, 0, 0, 0, 1)]
# family 4 (big tiles) exists only in a build with tools/diag/pointwise_big_tile.patch.txt applied
VARIANTS = [("tile32", (3, 0, 0)), ("128x128", (4, 1, 0)), ("64x128", (4, 2, 0)), ("64x192", (4, 3, 0)), ("64x128s", (4, 4, 0)), ("auto", (0, 0, 0))]
VARIANTS = [v for v in VARIANTS if _lib.load().ocv_pointwise_split_set_dispatch(*v[1]) == 0]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()
tot = {v: 0.0 for v, _ in VARIANTS}
print(f"{'layer':>22s} " + " ".join(f"{v:>9s}" for v, _ in VARIANTS) + "   (us per launch)")
for (M, Ci, Co, act, gate, res, n) in LAYERS:
    rpi = 1200 if M == M4 else 300
    x = torch.randn(M, Ci, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") * 0.05
    b = torch.randn(Co, device="cuda")
    g = torch.rand(M // rpi, Ci, device="cuda") if gate else None
    r = torch.randn(M, Co, device="cuda") if res else None
    sw = hip_ops.SplitWeight(w)
    x4 = x.view(M // rpi, rpi, 1, Ci).permute(0, 3, 1, 2)          # [B, C, rpi, 1] channels_last view
    r4 = None if r is None else r.view(M // rpi, rpi, 1, Co).permute(0, 3, 1, 2)
    row = []
    for name, cfg in VARIANTS:
        lib.ocv_pointwise_split_set_dispatch(*cfg)
        for _ in range(3):
            hip_ops.pointwise_nhwc(x4, sw, b, act, gate=g, residual=r4)
        torch.cuda.synchronize()
        hip_ops.enable_timing(True)
        for _ in range(reps):
            hip_ops.pointwise_nhwc(x4, sw, b, act, gate=g, residual=r4)
        us = list(hip_ops.timing_results().values())[0][1] * 1e3
        hip_ops.enable_timing(False)
        row.append(us)
        tot[name] += us * n
    print(f"{M:6d} {Ci:5d}->{Co:5d} x{n} " + " ".join(f"{u:9.1f}" for u in row))
lib.ocv_pointwise_split_set_dispatch(0, 0, 0)
print(f"{'per step (ms)':>22s} " + " ".join(f"{tot[v] / 1e3:9.3f}" for v, _ in VARIANTS))
