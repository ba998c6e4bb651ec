#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace23
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/gfwd_b1.py > $O/out.log 2>&1; echo "rc=$?"
tail -2 $O/out.log
cd $R
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:56]
    print("%8.1f %6.1f gap %5.1f q%s g%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Queue_Id"), r["Grid_Size_X"], n))
    prev = max(prev, e)
PY
rm -rf $O/t
