#!/usr/bin/env python3
"""Condense tools/profile_configs.sh output into profiles/<tag>_configs.json (one entry per BASELINE configuration: the
bench.py JSON line minus the bulky dtype note) and profiles/<tag>_cfg<c>_step_breakdown.txt (rocprofv3 kernel trace, fastest
step, one row per launch grid).

    python tools/make_configs_profile.py r03
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import summarize  # noqa: E402


def main(tag):
    out = {}
    for c in (1, 2, 3, 4):
        p = os.path.join(ROOT, "gpurun_out", "configs", f"cfg{c}.json")
        if not os.path.exists(p):
            continue
        line = [ln for ln in open(p).read().splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        d.pop("dtype_note", None)
        out[f"configs[{c}]"] = d
        src = os.path.join(ROOT, "gpurun_out", f"prof_cfg{c}")
        if os.path.isdir(os.path.join(src, "kt")):
            summarize.main(src, f"{tag}_cfg{c}")
            ks = os.path.join(ROOT, "profiles", f"{tag}_cfg{c}_kernel_stats.csv")
            if os.path.exists(ks) and c != 2:
                os.remove(ks)                      # the step breakdown is the summary kept per configuration
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_configs.json"), "w"), indent=1)
    for k, d in out.items():
        r = d.get("roofline") or {}
        print(f"{k}: {d['value']} img/s pipelined, {d.get('value_sequential')} sequential, sustained {d.get('sustained_images_per_s')}, "
              f"exact-fp32 {d.get('exact_fp32_images_per_s')}, cpu {d.get('cpu_baseline', {}).get('value')}, max_rel_vs_cpu {d.get('max_rel_vs_cpu')}, "
              f"roofline {r.get('kernel')} frac {r.get('frac')}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
