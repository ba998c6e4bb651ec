#!/bin/bash
# kernel stats of the RealMelGan variant's train step (graph replay, scales serialised)
export TMPDIR=/tmp MSYNTH_STREAMS=0
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_real
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_real -- python3 $R/bench.py --prime 0 --model realmelgan --steps 10 --warmup 6 > $R/gpurun_out/prof_real.json 2> $R/gpurun_out/prof_real.log
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_real/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms: %.2f" % (tot / 1e6))
for r in rows[:28]:
    print("%-96s %6s %9.2fms avg %8.1fus %5.1f%%" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
