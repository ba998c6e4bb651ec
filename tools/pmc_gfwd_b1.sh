#!/bin/bash
# Counters of the kernels of the B = 1 generator forward (BASELINE config 2; tools/gfwd_b1.py: three eager forwards + 200 graph
# replays): wave / wait cycles, instruction mix, LDS conflicts, and the bytes each launch fetches from beyond its L2
# (TCC FETCH_SIZE, x2 on gfx950 as in tools/pmc_summarize.py).  Separate --pmc passes, kernel-trace only.
# usage (GPU box): bash tools/pmc_gfwd_b1.sh [tag]     Output: gpurun_out/<tag>_pmc_gfwd_b1.txt
tag=${1:-r05}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_gfwd_b1/p$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_gfwd_b1/p$i -- python3 $R/tools/gfwd_b1.py > $R/gpurun_out/pmc_gfwd_b1.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_gfwd_b1.p$i.log; }
  echo "pass $i done"
done
cd $R
python3 - <<'PY' > gpurun_out/${tag}_pmc_gfwd_b1.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gfwd_b1/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void \(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch means over the run (tools/pmc_gfwd_b1.sh); FETCH_SIZE in KiB x 2 (gfx950) = bytes from beyond the L2")
for k, cs in sorted(agg.items()):
    if not k.startswith("k_"): continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    line = "%-62s launches %5d" % (k[:62], n)
    if "FETCH_SIZE" in m: line += "  fetch %8.2f MB" % (m["FETCH_SIZE"] * 2 * 1024 / 1e6)
    if wc:
        line += "  waves %6.0f  wait_any %4.0f%%  wait_inst %4.0f%%  active %4.0f%%" % (
            m.get("SQ_WAVES", 0), 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc)
    if "SQ_ACTIVE_INST_VALU" in m and wc:
        line += "  valu %4.0f%% lds %4.0f%% vmem %4.0f%%" % (100 * m["SQ_ACTIVE_INST_VALU"] / wc, 100 * m["SQ_ACTIVE_INST_LDS"] / wc, 100 * m["SQ_ACTIVE_INST_VMEM"] / wc)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        line += "  lds_conflict %4.0f%%" % (100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"])
    print(line)
PY
cat gpurun_out/${tag}_pmc_gfwd_b1.txt
