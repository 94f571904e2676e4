"""Late-stage 1x1 layers (encoder stages 4-7) at bs = 16: the round-2 route (fp32 rows, gate applied by the consumer:
pw_tile_kernel) against the pre-split route (hl32 rows by LDS-DMA, gate folded into per-image weights: pw_hl_kernel) on every
wavefront tile shape, plus the cost of the pieces the new route adds (split output of the depthwise kernel, gate + weights).
Usage: python tools/run_pw_hl.py [reps]      (table -> profiles/r03_pointwise_hl_sweep.txt)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objcavit_amd import hip_ops, _lib

M4, M6 = 19200, 4800
# (M, Cin, Cout, act, gate, res, launches per step)
LAYERS = [(M4, 384, 128, 0, 1, 0, 1), (M4, 128, 768, 3, 0, 0, 7), (M4, 768, 128, 0, 1, 1, 6), (M4, 768, 176, 0, 1, 0, 1),
          (M4, 176, 1056, 3, 0, 0, 7), (M4, 1056, 176, 0, 1, 1, 6), (M6, 1056, 304, 0, 1, 0, 1), (M6, 304, 1824, 3, 0, 0, 9),
          (M6, 1824, 304, 0, 1, 1, 8), (M6, 1824, 512, 0, 1, 0, 1), (M6, 512, 3072, 3, 0, 0, 2), (M6, 3072, 512, 0, 1, 1, 2)]
CFGS = [(1, 1), (1, 2), (2, 1), (2, 2), (4, 1), (4, 2), (0, 0)]
PANELS = ["1,1,0", "1,2,0", "2,1,0", "2,2,0", "1,1,1", "1,1,2", "1,1,4", "1,1,8", "1,2,1", "1,2,2", "1,2,4"]
panel_rows = []
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _lib.load()


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hip_ops.enable_timing(True)
    for _ in range(reps):
        fn()
    t = hip_ops.timing_results()
    hip_ops.enable_timing(False)
    return sum(v[1] for v in t.values()) * 1e3          # us per call (all launches of the call)


tot_old, tot_new = 0.0, {c: 0.0 for c in CFGS}
print(f"{'layer':>24s} {'tile32':>8s} | " + " ".join(f"{f'hl {r}x{t}':>8s}" for r, t in CFGS[:-1]) + f" {'hl auto':>8s}   (us per launch)")
for (M, Ci, Co, act, gate, res, n) in LAYERS:
    rpi = 1200 if M == M4 else 300
    B = M // rpi
    x = torch.randn(M, Ci, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") * 0.05
    b = torch.randn(Co, device="cuda")
    g = torch.rand(B, Ci, device="cuda") if gate else None
    r = torch.randn(M, Co, device="cuda") if res else None
    sw = hip_ops.SplitWeight(w)
    x4 = x.view(B, rpi, 1, Ci).permute(0, 3, 1, 2)          # [B, C, rpi, 1] channels_last view
    r4 = None if r is None else r.view(B, rpi, 1, Co).permute(0, 3, 1, 2)
    old = timeit(lambda: hip_ops.pointwise_nhwc(x4, sw, b, act, gate=g, residual=r4))
    xs = hip_ops.split_act(x4.contiguous(memory_format=torch.channels_last))
    if gate:
        img = int(lib.ocv_pointwise_packed_weight_elems(Ci, Co))
        packed = torch.cat([hip_ops.SplitWeight(w * g[i][None, :]).packed for i in range(B)])
        wq = hip_ops.PerImageSplitWeight(packed, Co, Ci, img, B)
    else:
        wq = sw
    row = []

    for cfg in CFGS:
        lib.ocv_pointwise_hl_set_dispatch(*cfg)
        us = timeit(lambda: hip_ops.pointwise_hl(xs, wq, b, act, residual=r4, out_fp32=True, out_split=(Co % 8 == 0 and not act)))
        row.append(us)
        tot_new[cfg] += us * n
    lib.ocv_pointwise_hl_set_dispatch(0, 0)
    if Co >= 4 * Ci and os.environ.get("OCV_RUN_PANEL"):   # expand layers: the row-panel form (tools/diag/pointwise_row_panel.patch.txt applied), "rt,tn,nsplit"
        prow = []
        for pc in PANELS:
            os.environ["OCV_PWHL_PANEL"] = pc
            prow.append(timeit(lambda: hip_ops.pointwise_hl(xs, wq, b, act, residual=r4, out_fp32=True, out_split=False)))
        os.environ.pop("OCV_PWHL_PANEL")
        panel_rows.append((M, Ci, Co, n, prow))
    tot_old += old * n
    print(f"{M:6d} {Ci:5d}->{Co:5d} x{n} {'gate' if gate else '    '} {old:8.1f} | " + " ".join(f"{u:8.1f}" for u in row))
print(f"{'per step (ms)':>24s} {tot_old / 1e3:8.3f} | " + " ".join(f"{tot_new[c] / 1e3:8.3f}" for c in CFGS))

print("\nexpand layers, row-panel form (OCV_PWHL_PANEL = rt,tn,nsplit; nsplit 0 = automatic):")
print(f"{'layer':>24s} " + " ".join(f"{pc:>8s}" for pc in PANELS))
for (M, Ci, Co, n, prow) in panel_rows:
    print(f"{M:6d} {Ci:5d}->{Co:5d} x{n}      " + " ".join(f"{u:8.1f}" for u in prow))

# the pieces the pre-split route adds / replaces around the GEMMs
print("\ndepthwise + squeeze-excite tail: fp32 output + gate (round 2) vs hl32 output + gate folded into per-image project weights")
for (Bn, C, H, W, k, s, R, N) in [(16, 768, 30, 40, 3, 1, 32, 128), (16, 1056, 30, 40, 5, 1, 44, 176), (16, 1824, 15, 20, 5, 1, 76, 304),
                                  (16, 3072, 15, 20, 3, 1, 128, 512)]:
    x = torch.randn(Bn, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    wd = torch.randn(k * k, C, device="cuda") * 0.2
    bd = torch.randn(C, device="cuda") * 0.1
    w1, b1 = torch.randn(R, C, device="cuda") * 0.03, torch.randn(R, device="cuda") * 0.1
    w2t, b2 = torch.randn(R, C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    wp = torch.randn(N, C, device="cuda") * 0.03
    a = timeit(lambda: hip_ops.depthwise_se_gate(x, wd, bd, k, s, w1, b1, w2t, b2))
    c = timeit(lambda: hip_ops.depthwise_se_gate_weights(x, wd, bd, k, s, w1, b1, w2t, b2, wp))
    print(f"  B{Bn} {H}x{W} C={C} k{k}: fp32 + gate {a:7.1f} us   hl32 + gate-weights (N={N}) {c:7.1f} us")
