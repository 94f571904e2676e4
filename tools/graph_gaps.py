"""Where does a slow replay spend its time?  python tools/graph_gaps.py <rocprofv3 -d dir> : the LAST step of the kernel trace
(launches between two bin_head kernels), its wall time, the sum of kernel time, and the largest idle gaps between consecutive
launches (by start time) with the kernels on either side and their queue ids."""
import csv, glob, os, sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "bin_head" in r["Kernel_Name"]]
step = rows[ends[-2] + 1: ends[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"last step: {len(step)} launches, wall {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, queues {sorted(set(r['Queue_Id'] for r in step))}")
gaps = []
hi = t0
for a, b in zip(step[:-1], step[1:]):
    hi = max(hi, int(a["End_Timestamp"]))
    gaps.append((int(b["Start_Timestamp"]) - hi, a, b))
short = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
    print(f"  gap {g / 1e3:8.1f} us  after {short(a)} [q{a['Queue_Id']}]  before {short(b)} [q{b['Queue_Id']}]  at t = {(int(b['Start_Timestamp']) - t0) / 1e3:.1f} us")
print(f"  sum of positive gaps {sum(max(0, g) for g, _, _ in gaps) / 1e3:.1f} us; gaps > 20 us: {sum(1 for g, _, _ in gaps if g > 20000)}")
