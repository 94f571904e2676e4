#!/bin/bash
# rocprofv3 kernel trace of tools/exp_wino43_parts.py per shape -> gpurun_out/wino43_parts/<shape>.txt (per-kernel average durations)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wino43_parts
for shape in "16 240 320 128 128" "16 120 160 256 256" "16 60 80 512 512"; do
  tag=$(echo $shape | tr ' ' '_')
  rm -rf gpurun_out/wino43_parts/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wino43_parts/$tag -- python3 tools/exp_wino43_parts.py $shape > gpurun_out/wino43_parts/$tag.log 2>&1 || { tail -5 gpurun_out/wino43_parts/$tag.log; exit 1; }
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/wino43_parts/{tag}/*/*_kernel_trace.csv"))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("wino43", "conv_split_dma")):
        g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else 0
        d[(n.split("(")[0][-40:], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"gpurun_out/wino43_parts/{tag}.txt", "w") as o:
    for (n, g), v in sorted(d.items()):
        v = sorted(v)[: max(1, len(v) - 2)]                    # drop the two slowest (first calls)
        o.write(f"{tag}  {n:42s} wgs={g:6d}  n={len(v)}  avg {sum(v) / len(v) / 1e3:8.1f} us\n")
print(open(f"gpurun_out/wino43_parts/{tag}.txt").read())
PY
done
