#!/bin/bash
# per-kernel durations of the few-key cross-attention call at several batches (rocprofv3 --kernel-trace --stats on tools/xattn_one.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${1:-300}
for B in 16 64 128 512 2048; do
  d=gpurun_out/xtrace_$B; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/xattn_one.py $B $S > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
  python3 - $d $B <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "xattn" in r["Name"]:
        print(f"B={sys.argv[2]:>5s} {r['Name'][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
PY
  rm -rf $d $d.log
done
