#!/bin/bash
# per-kernel time of a do_final_upscale GraphBins forward (eager, bs = $1, default 16): rocprofv3 --kernel-trace --stats on
# tools/run_final_upscale.py -> gpurun_out/final_upscale_kernels.txt (top kernels by total time, per forward)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-16}
d=gpurun_out/prof_fu; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/run_final_upscale.py $B > gpurun_out/final_upscale_run.log 2>&1 || { tail -5 gpurun_out/final_upscale_run.log; exit 1; }
tail -n 1 gpurun_out/final_upscale_run.log
python3 - $d <<'PY' > gpurun_out/final_upscale_kernels.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
n_fwd = 13                                   # 3 warm-up + 10 timed forwards
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU time per forward: {tot / n_fwd / 1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{float(r['TotalDurationNs']) / n_fwd / 1e6:8.3f} ms  {float(r['Calls']) / n_fwd:6.1f} calls  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")
PY
python3 - $d <<'PY' >> gpurun_out/final_upscale_kernels.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_split_dma", "tap_interp", "bin_head", "conv_exact", "wino"))]
per = len(big) // 13
print(f"\nlast forward, the convolution / interpolation / bin-head launches in order ({per} per forward): us, grid, kernel")
for r in big[-per:]:
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f}  grid {r['Grid_Size_X']:>9s} x {r['Grid_Size_Y']:>5s} x {r['Grid_Size_Z']:>3s}  {r['Kernel_Name'][:70]}")
PY
rm -rf $d
cat gpurun_out/final_upscale_kernels.txt
