"""Per-launch timeline of one bench step: duration (kernel-trace pass) beside the HBM bytes of the same launch (separate --pmc
passes of the same command: 2 x FETCH_SIZE + WRITE_SIZE KB, the gfx950 correction of MI355X_MICROARCH.md), in dispatch order.
python tools/step_timeline.py gpurun_out/prof > profiles/<tag>_timeline.txt"""
import csv, glob, os, sys

def newest(p):
    return sorted(glob.glob(p), key=os.path.getmtime)[-1]

def is_end(n):
    return "bin_head" in n

def steps(rows, name_key):
    out, cur = [], []
    for r in rows:
        cur.append(r)
        if is_end(r[name_key]):
            out.append(cur)
            cur = []
    return out

src = sys.argv[1]
kt = sorted(csv.DictReader(open(newest(os.path.join(src, "kt", "*", "*_kernel_trace.csv")))), key=lambda r: int(r["Dispatch_Id"]))
ks = steps(kt, "Kernel_Name")
n = max(set(len(s) for s in ks[2:]), key=[len(s) for s in ks[2:]].count)
cand = [s for s in ks[2:] if len(s) == n]
best = min(cand, key=lambda s: int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]))

def pmc(d, counter):
    if not glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
        return None
    rows = [r for r in csv.DictReader(open(newest(os.path.join(src, d, "*", "*_counter_collection.csv")))) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    st = [s for s in steps(rows, "Kernel_Name") if len(s) == n]
    return st[-1] if st else None

f, w = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
t0 = int(best[0]["Start_Timestamp"])
print(f"# {n} launches, wall {(int(best[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
print(f"{'#':>3} {'t_us':>8} {'dur_us':>8} {'gap_us':>6} {'fetch_MB':>9} {'write_MB':>9} {'GB/s':>7}  kernel [grid x block]")
prev = t0
for i, r in enumerate(best):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    fb = 2 * float(f[i]["Counter_Value"]) * 1024 if f and f[i]["Kernel_Name"] == r["Kernel_Name"] else float("nan")
    wb = float(w[i]["Counter_Value"]) * 1024 if w and w[i]["Kernel_Name"] == r["Kernel_Name"] else float("nan")
    g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    print(f"{i:3d} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:6.1f} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {(fb + wb) / (e - s):7.0f}  {name} [{g}x{r['Workgroup_Size_X']}]")
    prev = e
