"""Why do the encoder's 1x1 launches take 1.5 - 2 x their microbenchmark time inside the step (profiles/r05_pw_tile_isa.txt)?  Each layer timed
(HIP events around the launch alone) three ways: `hot` = the same input buffer every launch (what tools/history/run_pw.py and run_pw_late.py
did), `fresh` = the input rewritten by an element-wise kernel right before each launch (a producer, as in the step), `rotate` = 24 different
input / output buffers in turn, each rewritten before its launch (nothing of the previous launch's working set helps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops
M4, M6 = 19200, 4800
LAYERS = [(M4, 128, 768, 3, 0, 0), (M4, 768, 128, 0, 1, 1), (M4, 176, 1056, 3, 0, 0), (M4, 1056, 176, 0, 1, 1), (M6, 304, 1824, 3, 0, 0),
          (M6, 1824, 304, 0, 1, 1), (M6, 512, 3072, 3, 0, 0), (M6, 3072, 512, 0, 1, 1)]
reps, NB = 48, 24
print(f"{'layer':>20s} {'hot':>8s} {'fresh':>8s} {'rotate':>8s}   (us per launch)")
for (M, Ci, Co, act, gate, res) in LAYERS:
    rpi = 1200 if M == M4 else 300
    xs = [torch.randn(M, Ci, device="cuda") for _ in range(NB)]
    w = torch.randn(Co, Ci, device="cuda") * 0.05
    b = torch.randn(Co, device="cuda")
    g = torch.rand(M // rpi, Ci, device="cuda") if gate else None
    rs = [torch.randn(M, Co, device="cuda") if res else None for _ in range(NB)]
    sw = hip_ops.SplitWeight(w)
    v4 = lambda t, C: None if t is None else t.view(M // rpi, rpi, 1, C).permute(0, 3, 1, 2)   # noqa: E731
    out = []
    for mode in ("hot", "fresh", "rotate"):
        for _ in range(3):
            hip_ops.pointwise_nhwc(v4(xs[0], Ci), sw, b, act, gate=g, residual=v4(rs[0], Co))
        torch.cuda.synchronize()
        hip_ops.enable_timing(True)
        for i in range(reps):
            j = i % NB if mode == "rotate" else 0
            if mode != "hot":
                xs[j].mul_(1.0)                                   # the producer: rewrites the input right before the launch
            hip_ops.pointwise_nhwc(v4(xs[j], Ci), sw, b, act, gate=g, residual=v4(rs[j], Co))
        out.append(list(hip_ops.timing_results().values())[0][1] * 1e3)
        hip_ops.enable_timing(False)
    print(f"{M:6d} {Ci:5d}->{Co:5d} " + " ".join(f"{u:8.1f}" for u in out))
