"""Condense tools/pmc_binhead.sh: per variant, the counters of the bs-16 launches (grid 256 x 512) of the bin head kernel, per launch."""
import collections, csv, glob, os, sys
src = sys.argv[1]
for v in sorted({os.path.basename(d).rsplit("_", 1)[0] for d in glob.glob(os.path.join(src, "*_[abc]"))}):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    dur = []
    for p in "abc":
        for f in glob.glob(os.path.join(src, f"{v}_{p}", "*", "*_counter_collection.csv")):
            rows = [r for r in csv.DictReader(open(f)) if "bin_head_h2" in r["Kernel_Name"]]
            first = sorted({int(r["Dispatch_Id"]) for r in rows})[:30]          # the tool's first size is bs 16 (5 + 30 + 1 launches)
            for r in rows:
                if int(r["Dispatch_Id"]) in first:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
                    n[r["Counter_Name"]].add(r["Dispatch_Id"])
                    if p == "a" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
                        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"== {v}: {len(dur)} launches at bs 16, mean {sum(dur) / max(len(dur), 1) / 1e3:.1f} us under counters")
    for k in sorted(tot):
        print(f"   {k:32s} {tot[k] / len(n[k]):16.0f} per launch")
    w = tot["SQ_WAVE_CYCLES"] / max(len(n["SQ_WAVE_CYCLES"]), 1)
    if w:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in tot:
                print(f"   {k} / SQ_WAVE_CYCLES = {tot[k] / len(n[k]) / w:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
            print(f"   matrix pipe busy / (4 x wave quad-cycles / 2 waves per SIMD) = {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / len(n['SQ_VALU_MFMA_BUSY_CYCLES']) / (4 * w / 2):.3f}")
