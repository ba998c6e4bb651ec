#!/bin/bash
# rocprofv3 kernel trace of the replayed train step -> gpurun_out/trace9/compact.csv (start, end, queue, kernel; ns) for
# tools/trace_timeline.py (alone / overlapped time per kernel family) and tools/trace_sequence.py (one call, in order).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace9
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/replay -- python3 $R/bench.py --prime 0 --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-gforward > $O/replay_bench.json 2> $O/replay_bench.log; echo "replay rc=$?"
cd $R
f=$(find $O/replay -name "*kernel_trace.csv" | head -1); ls -la $f
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
# keep the last ~8 steps worth: write a compact file: start, end, stream/queue, name
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("gpurun_out/trace9/compact.csv", "w") as f:
    for r in rows:
        f.write("%d,%d,%s,%s\n" % (int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", ""), r["Kernel_Name"][:70].replace(",", ";")))
PY
rm -rf $O/replay
ls -la gpurun_out/trace9
