"""Per-launch means of the counters collected by tools/pmc_pwhl.sh for the pointwise kernel of the run."""
import collections, csv, glob, os, sys
d = sys.argv[1]
agg = collections.defaultdict(list)
dur = []
for sub in "abcfw":
    fs = glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if "pw_hl_kernel" in r["Kernel_Name"] or "pw_panel_kernel" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            if sub == "a":
                per[int(r["Dispatch_Id"])]["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ids = sorted(per)[len(per) // 2:]                      # second half: warm
    for i in ids:
        for k, v in per[i].items():
            agg[k].append(v)
m = {k: sum(v) / len(v) for k, v in agg.items()}
ns = m.get("_ns", 0)
print(f"{d}: {ns / 1e3:.1f} us per launch under PMC")
for k in sorted(m):
    if k != "_ns":
        print(f"  {k:28s} {m[k]:16.0f}")
N_SIMD = 1024
if ns:
    cyc = ns * 2.1
    print(f"  mfma busy (of {N_SIMD} SIMDs x {cyc:.0f} cyc @2.1GHz): {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * N_SIMD):.3f}")
if m.get("SQ_INSTS_MFMA"):
    print(f"  VALU per MFMA {m.get('SQ_INSTS_VALU', 0) / m['SQ_INSTS_MFMA']:.2f}  SALU per MFMA {m.get('SQ_INSTS_SALU', 0) / m['SQ_INSTS_MFMA']:.2f}  "
          f"LDS per MFMA {m.get('SQ_INSTS_LDS', 0) / m['SQ_INSTS_MFMA']:.2f}  VMEM per MFMA {m.get('SQ_INSTS_VMEM', 0) / m['SQ_INSTS_MFMA']:.2f}")
if m.get("SQ_WAVE_CYCLES"):
    print(f"  wait_any/wave_cycles {m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}  wait_inst_any {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}  "
          f"wait_inst_lds {m.get('SQ_WAIT_INST_LDS', 0) / m['SQ_WAVE_CYCLES']:.3f}  active_inst_any {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}")
if m.get("SQ_BUSY_CYCLES"):
    print(f"  mfma_busy/sq_busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / m['SQ_BUSY_CYCLES']:.3f}  lds_bank_conflict/sq_busy {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_BUSY_CYCLES']:.3f}")
if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
    print(f"  traffic: 2 x FETCH_SIZE {2 * m.get('FETCH_SIZE', 0) / 1024:.1f} MB + WRITE_SIZE {m.get('WRITE_SIZE', 0) / 1024:.1f} MB")
