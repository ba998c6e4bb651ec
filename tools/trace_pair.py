"""Kernel sequence of the LAST D,G pair of a rocprofv3 --kernel-trace csv (ends at the last k_adam): index, start (us from the
pair's first kernel), duration (us), queue, name.  usage: trace_pair.py <dir or kernel_trace.csv>"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else max(glob.glob(os.path.join(p, "**/*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "k_adam(" in n]
lo, hi = adam[-3] + 1, adam[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0.0
for i in range(lo, hi):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += (e - s) / 1e3
    n = names[i].replace("void ", "").replace("(anonymous namespace)::", "")
    print("%3d %8.1f %7.1f q%s %s" % (i - lo, (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n[:110]))
print("kernels %d, sum of durations %.1f us, span %.1f us" % (hi - lo, tot, (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
