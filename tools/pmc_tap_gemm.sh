#!/bin/bash
# L2-side counters of the tap GEMM (conv_split_dma_kernel, 1 x 1, K = 256 -> 1152 at 120 x 160, bs 16): how many bytes the launch pulls
# through L2 into LDS against the 0.63 GB it reads from HBM.  One counter group per pass (rocprofv3 --pmc with --kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_tap_gemm
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/run_tap_gemm.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -1 $OUT/p$i.log | cut -c1-160)"
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
src = sys.argv[1]
tot, n, dur = collections.defaultdict(float), collections.defaultdict(set), []
for f in glob.glob(os.path.join(src, "p*", "*", "*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "conv_split_dma_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[2:]          # skip the first two launches (cold caches)
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
            if r["Counter_Name"] in ("FETCH_SIZE",): dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"== conv_split_dma_kernel (tap GEMM), per launch, {len(dur)} launches, {sum(dur) / max(len(dur), 1) / 1e3:.1f} us under the FETCH_SIZE pass")
for c in sorted(tot): print(f"   {c:28s} {tot[c] / len(n[c]):18.0f}")
PY
