#!/bin/bash
# Final-state evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the train-step bench alone (graph
# replay, streams on; the config-2 B=1 leg excluded so B=32 and B=1 dispatches do not share rows), (2) the same with
# the streams serialised (every kernel alone: its traced duration is its own), (3) the config-2 leg on its own,
# (4) the bench JSON line of the same code.  Everything lands under gpurun_out/final/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/replay -- python3 $R/bench.py --prime 0 --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --no-gforward --no-exact --no-dp-overhead > $O/replay_bench.json 2> $O/replay_bench.log; echo "replay rc=$?"
MSYNTH_STREAMS=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/bench.py --prime 0 --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --no-gforward --no-exact --no-dp-overhead > $O/serial_bench.json 2> $O/serial_bench.log; echo "serial rc=$?"
cd $R
timeout -k 10 600 python3 bench.py --steps 20 --warmup 6 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
python3 - <<'PY'
import csv, glob
for tag in ("replay", "serial"):
    fs = glob.glob("gpurun_out/final/%s/**/*kernel_stats.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no stats"); continue
    rows = list(csv.DictReader(open(fs[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("== %s: total kernel ms %.2f (%d kernels)" % (tag, tot / 1e6, len(rows)))
    for r in rows[:16]:
        print("%-92s %6s %9.2fms avg %8.1fus %5.1f%%" % (r["Name"][:92], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
