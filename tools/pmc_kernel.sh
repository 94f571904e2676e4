#!/bin/bash
# SQ counters of ONE kernel under a small driver script: tools/pmc_kernel.sh <kernel-name-substring> <script.py> [args]   (GPU box, through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=$1; shift
OUT=gpurun_out/pmc_kernel
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 "$@" > $OUT/a.log 2>&1 || { tail -3 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 "$@" > $OUT/b.log 2>&1 || { tail -3 $OUT/b.log; exit 1; }
python3 - "$K" $OUT <<'PY'
import collections, csv, glob, os, sys
k, src = sys.argv[1], sys.argv[2]
tot, n, dur = collections.defaultdict(float), collections.defaultdict(set), []
for p in "ab":
    for f in glob.glob(os.path.join(src, p, "*", "*_counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if k in r["Kernel_Name"]]
        first = sorted({int(r["Dispatch_Id"]) for r in rows})[:30]
        for r in rows:
            if int(r["Dispatch_Id"]) in first:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
                if p == "a" and r["Counter_Name"] == "SQ_WAVE_CYCLES": dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"== {k}: first {len(dur)} launches, mean {sum(dur) / max(len(dur), 1) / 1e3:.1f} us under counters")
for c in sorted(tot): print(f"   {c:28s} {tot[c] / len(n[c]):16.0f} per launch")
w = tot["SQ_WAVE_CYCLES"] / max(len(n["SQ_WAVE_CYCLES"]), 1)
for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
    if c in tot and w: print(f"   {c} / SQ_WAVE_CYCLES = {tot[c] / len(n[c]) / w:.3f}")
PY
