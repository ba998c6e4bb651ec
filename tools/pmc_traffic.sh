#!/bin/bash
# HBM traffic of every kernel from the TCC counters (separate --pmc passes, as the gfx950 guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; kernel-trace only).  Eager, stream-serialised
# train steps so every dispatch is attributed to one kernel.  Output: gpurun_out/pmc_{FETCH,WRITE}_SIZE/ and
# gpurun_out/r05_pmc_traffic.json = {"code_version": <bench.code_version()>, "kernels": {...}} -- copy it to
# profiles/ for the state it was taken at.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_$c
  rm -rf $d
  MSYNTH_STREAMS=0 MSYNTH_GRAPH=0 timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- \
      python3 $R/bench.py --prime 0 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-exact --no-dp-overhead --no-gforward > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.log
  echo "pmc $c rc=$?"
done
cd $R
python3 tools/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_kernels.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
k = json.load(open("gpurun_out/pmc_kernels.json"))
json.dump({"code_version": bench.code_version(), "dg_pairs": 5,       # 4 priming + 2 warm-up + 4 timed calls = 5 D+G pairs
           "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace on "
                             "bench.py --steps 4 --warmup 2, MSYNTH_STREAMS=0 MSYNTH_GRAPH=0; FETCH_SIZE x2 (gfx950), KiB units",
           "kernels": k}, open("gpurun_out/r05_pmc_traffic.json", "w"), indent=1)
tot = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in k.values())
print("code_version", bench.code_version(), "kernels", len(k), "total HBM GB over the run: %.2f" % (tot / 1e9))
PY
