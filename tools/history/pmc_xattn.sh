#!/bin/bash
# SQ counters of the few-key cross-attention kernels at a large batch (tools/xattn_one.py B S; OCV_XATTN_FORM = h2 | split3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/pmc_xattn; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $d/a -- python3 tools/xattn_one.py "$@" > $d/a.log 2>&1 || { tail -3 $d/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $d/b -- python3 tools/xattn_one.py "$@" > $d/b.log 2>&1 || { tail -3 $d/b.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH --output-format csv -d $d/c -- python3 tools/xattn_one.py "$@" > $d/c.log 2>&1 || { tail -3 $d/c.log; }
python3 - $d <<'PY'
import collections, csv, glob, sys
d = sys.argv[1]
for kern in ("xattn_main_h2_kernel", "xattn_kv_h2_kernel", "xattn_main3_kernel", "xattn_kv3_kernel"):
    m = collections.defaultdict(list); ns = []
    for sub in "abc":
        fs = glob.glob(f"{d}/{sub}/*/*_counter_collection.csv")
        if not fs: continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(fs[0])):
            if kern in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                if sub == "a": per[int(r["Dispatch_Id"])]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for i in sorted(per)[len(per) // 2:]:
            for k, v in per[i].items(): m[k].append(v)
    a = {k: sum(v) / len(v) for k, v in m.items()}
    if not a: continue
    cyc = a.get("_ns", 0) * 2.1
    print(f"{kern}: {a.get('_ns', 0) / 1e3:.1f} us; waves {a.get('SQ_WAVES', 0):.0f}; per wave: MFMA {a.get('SQ_INSTS_MFMA', 0) / max(a.get('SQ_WAVES', 1), 1):.0f} VALU {a.get('SQ_INSTS_VALU', 0) / max(a.get('SQ_WAVES', 1), 1):.0f} "
          f"SALU {a.get('SQ_INSTS_SALU', 0) / max(a.get('SQ_WAVES', 1), 1):.0f} LDS {a.get('SQ_INSTS_LDS', 0) / max(a.get('SQ_WAVES', 1), 1):.0f} VMEM {a.get('SQ_INSTS_VMEM', 0) / max(a.get('SQ_WAVES', 1), 1):.0f}; "
          f"wave cycles per wave {a.get('SQ_WAVE_CYCLES', 0) * 4 / max(a.get('SQ_WAVES', 1), 1):.0f} (x4: quad-cycles); mfma busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(cyc * 1024, 1):.3f}; "
          f"wait_any {a.get('SQ_WAIT_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.3f} wait_inst_any {a.get('SQ_WAIT_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.3f} wait_lds {a.get('SQ_WAIT_INST_LDS', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.3f} "
          f"active_any {a.get('SQ_ACTIVE_INST_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.3f} active_valu {a.get('SQ_ACTIVE_INST_VALU', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.3f} active_lds {a.get('SQ_ACTIVE_INST_LDS', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):.3f} "
          f"lds_conflict/lds_active {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f} conflict/busy {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_BUSY_CYCLES', 1), 1):.3f}")
PY
