#!/bin/bash
# Kernel trace of the graph-replayed train step with the discriminator scales serialised
# (MSYNTH_STREAMS=0): every kernel runs alone, so its traced duration is its own.
export TMPDIR=/tmp MSYNTH_STREAMS=${MSYNTH_STREAMS:-0}
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_serial
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --prime 0 --steps 10 --warmup 6 --no-cpu-baseline --no-roofline --no-exact --no-dp-overhead --no-gforward > $R/gpurun_out/prof_serial.json 2> $R/gpurun_out/prof_serial.log
echo "rc=$?"; cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_serial/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms: %.2f" % (tot / 1e6))
for r in rows[:100]:
    print("%-100s %6s %9.2fms avg %8.1fus %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
python3 tools/trace_pair.py gpurun_out/prof_serial > gpurun_out/prof_serial_pair.txt
