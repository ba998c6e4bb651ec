#!/bin/bash
# Which kernels does ANYTHING dispatch?  rocprofv3 --kernel-trace --stats of (1) the whole GPU test suite, (2) the three bench
# models.  Output: gpurun_out/dispatch/{tests,headline,twostage,realmelgan}/**/*kernel_stats.csv and
# gpurun_out/dispatch/kernels.txt = the union of kernel base names (template arguments stripped).  A kernel that is in none
# of them is dead code for every exercised path (tools/dead_kernels.py lists them against csrc/*.hip).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dispatch
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tests -- python3 -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
for m in headline twostage realmelgan; do
  extra=""; [ $m != headline ] && extra="--model $m"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python3 bench.py $extra --prime 0 --steps 4 --warmup 2 --no-cpu-baseline --no-exact --no-dp-overhead > $O/$m.json 2> $O/$m.log; echo "$m rc=$?"
done
python3 - <<'PY'
import csv, glob, re
names = set()
for f in glob.glob("gpurun_out/dispatch/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(k_\w+)", r["Name"])
        if m: names.add(m.group(1))
open("gpurun_out/dispatch/kernels.txt", "w").write("\n".join(sorted(names)) + "\n")
print(len(names), "kernel base names dispatched")
PY
