#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== build" && python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q ${XFLAG--x} --timeout=600 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n ${TAILN:-40} gpurun_out/pytest_gpu.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && [ -z "$KEEP_GOING" ] && exit $rc
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; tail -5 gpurun_out/smoke.log
[ $rc -ne 0 ] && [ -z "$KEEP_GOING" ] && exit $rc
echo "== bench"
timeout -k 10 600 python bench.py --steps ${STEPS:-20} --warmup 6 > gpurun_out/bench.json 2> gpurun_out/bench.log; rc=$?
tail -25 gpurun_out/bench.log; cat gpurun_out/bench.json
[ $rc -ne 0 ] && exit $rc
if [ -n "$PROFILE" ]; then
  echo "== rocprofv3"
  cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --prime 0 --steps 10 --warmup 6 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log
  echo "rocprof rc=$?"; cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head
fi
