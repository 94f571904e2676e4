#!/bin/bash
# HBM traffic of the few-key cross-attention call at one batch (tools/xattn_one.py B S), FETCH_SIZE and WRITE_SIZE in rocprofv3
# passes of their own (no trace domain but --kernel-trace), corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE
# tallies 128-byte requests at 64 bytes -> x 2; both counters are in KB), per launch, against the algorithmic bytes of SURVEY 8d.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-512}; S=${2:-300}
d=gpurun_out/pmc_xt; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/f -- python3 tools/xattn_one.py $B $S > $d/f.log 2>&1 || { tail -3 $d/f.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $d/w -- python3 tools/xattn_one.py $B $S > $d/w.log 2>&1 || { tail -3 $d/w.log; exit 1; }
python3 - $d $B $S <<'PY'
import collections, csv, glob, sys
d, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
alg = B * (3 * S * 128 * 4 + S) + 4 * 128 * 128 * 4 + 4 * 128 * 4
tot = {}
for sub, cname, mul in (("f", "FETCH_SIZE", 2.0), ("w", "WRITE_SIZE", 1.0)):
    f = glob.glob(f"{d}/{sub}/*/*_counter_collection.csv")[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "xattn" in r["Kernel_Name"] and r["Counter_Name"] == cname:
            name = next(k for k in ("xattn_kv_h2_kernel", "xattn_main_h2_kernel", "xattn_kv3_kernel", "xattn_main3_kernel", "xattn") if k in r["Kernel_Name"])
            per[name][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in per.items():
        vals = [v[i] for i in sorted(v)][len(v) // 2:]          # the later launches (warm)
        tot.setdefault(k, {})[cname] = mul * 1024 * sum(vals) / len(vals)
s = 0.0
for k, v in tot.items():
    print(f"{k:42s} fetch {v.get('FETCH_SIZE', 0) / 1e6:8.1f} MB  write {v.get('WRITE_SIZE', 0) / 1e6:8.1f} MB per launch")
    s += v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)
print(f"B={B} S={S}: HBM traffic of the call {s / 1e6:.1f} MB (PMC) against {alg / 1e6:.1f} MB algorithmic = {s / alg:.2f}x")
PY
rm -rf $d
