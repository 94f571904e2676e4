#!/usr/bin/env python3
"""Condense tools/profile_smallbatch.sh output into profiles/<tag>_smallbatch.txt (+ one step breakdown per batch):

    python tools/make_smallbatch_profile.py r04a [gpurun_out/smallbatch]

Per batch: images/s and step latency of `bench.py --batch b --inflight 1`, and from the rocprofv3 kernel trace of the same
command the launches per step, the sum of kernel time, and the share of the step spent in launches shorter than 10 us.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import summarize  # noqa: E402


def step_of(trace):
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if summarize.is_step_end(r["Kernel_Name"])]
    steps = [rows[a + 1: b + 1] for a, b in zip(idx[:-1], idx[1:])]
    # the metric launches of step i trail its bin head: a "step" here = everything between two bin-head launches
    return min(steps, key=lambda st: int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])), len(steps)


def main(tag, src):
    out = os.path.join(ROOT, "profiles")
    lines = [f"# {tag}: the hot path at the batches the reference's validation loop runs (main.py:58: bs 1; GraphBinsLM.py:159,173: image + mirror = 2)",
             "# bench.py --batch b --inflight 1 --steps 40 (hipGraph replay, one batch after the other) + rocprofv3 --kernel-trace of the same command",
             f"{'bs':>3s} {'img/s':>8s} {'ms/step':>8s} {'launches':>8s} {'kernel_ms':>9s} {'wall_ms':>8s} {'<10us n':>8s} {'<10us ms':>9s} {'<10us %':>8s} {'<20us %':>8s}"]
    for f in sorted(glob.glob(os.path.join(src, "bs*.json")), key=lambda p: int(os.path.basename(p)[2:-5])):
        b = int(os.path.basename(f)[2:-5])
        try:
            res = json.loads(open(f).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            continue
        kt = sorted(glob.glob(os.path.join(src, f"kt_bs{b}", "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1:]
        n = ksum = wall = n10 = t10 = t20 = 0
        if kt:
            step, _ = step_of(kt[0])
            durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step]
            n, ksum = len(step), sum(durs) / 1e6
            wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
            n10, t10 = sum(1 for d in durs if d < 10000), sum(d for d in durs if d < 10000) / 1e6
            t20 = sum(d for d in durs if d < 20000) / 1e6
            agg = collections.defaultdict(lambda: [0, 0])
            for r in step:
                a = agg[summarize.shape_key(r)]
                a[0] += 1
                a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            with open(os.path.join(out, f"{tag}_bs{b}_step_breakdown.txt"), "w") as g:
                g.write(f"# bs {b}: fastest bench step in {os.path.basename(kt[0])}: wall {wall:.3f} ms, {n} kernels, sum of kernel time {ksum:.3f} ms\n")
                g.write(f"{'kernel':110s} {'n':>5s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}\n")
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    g.write(f"{k:110s} {v[0]:5d} {v[1] / 1e6:9.3f} {v[1] / v[0] / 1e3:9.1f} {100 * v[1] / (ksum * 1e6):6.2f}\n")
        lines.append(f"{b:3d} {res['value']:8.1f} {res['ms_per_step']:8.3f} {n:8d} {ksum:9.3f} {wall:8.3f} {n10:8d} {t10:9.3f} "
                     f"{100 * t10 / max(ksum, 1e-9):8.1f} {100 * t20 / max(ksum, 1e-9):8.1f}")
    txt = "\n".join(lines) + "\n"
    open(os.path.join(out, f"{tag}_smallbatch.txt"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "smallbatch"))
