#!/bin/bash
# LDS counters of the tap interpolation kernel for one library build: tools/pmc_tap.sh TAG [lib.so]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; d=gpurun_out/pmc_tap/$tag; rm -rf $d; mkdir -p $d
[ -n "$2" ] && export OCV_LIB_PATH=$2
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $d/a -- python3 tools/tap_ab.py > $d/a.log 2>&1 || { tail -3 $d/a.log; exit 1; }
python3 - $d <<'PY'
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/a/*/*_counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "tap_interp_kernel" in r["Kernel_Name"]:
        g = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        per[g][r["Counter_Name"]] += float(r["Counter_Value"])
        per[g]["_n"] += 1.0 / 8
for g, m in sorted(per.items()):
    print(f"{d} grid {g:6d}: launches {m['_n']:.0f}  LDS_BANK_CONFLICT / SQ_BUSY {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_BUSY_CYCLES']:.3f}  "
          f"CONFLICT / LDS_IDX_ACTIVE {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.3f}  LDS insts per wave-cycle {m['SQ_INSTS_LDS'] / m['SQ_WAVE_CYCLES']:.4f}  "
          f"VALU {m['SQ_INSTS_VALU'] / m['_n']:.0f}  wait_lds/wave_cycles {m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES']:.3f}")
PY
