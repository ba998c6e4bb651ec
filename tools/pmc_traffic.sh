#!/bin/bash
# HBM traffic of every kernel from the TCC counters (separate --pmc passes, as the gfx950 guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output: gpurun_out/pmc_{fetch,write}/
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$c
  rm -rf $d
  MSYNTH_STREAMS=0 MSYNTH_GRAPH=0 timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.log
  echo "pmc $c rc=$?"
done
cd $R
python3 tools/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
head -c 1500 gpurun_out/pmc_traffic.json
