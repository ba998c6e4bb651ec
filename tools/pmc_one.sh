#!/bin/bash
# usage: tools/pmc_one.sh <tag> <args to pmc_one.py...>   (separate --pmc passes, kernel-trace only)
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python3 $R/tools/pmc_one.py "$@" > $R/gpurun_out/pmc_$tag.p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_$tag.p$i.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$tag/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not any(t in k for t in ("gconv", "reduce", "wgrad", "rows", "atom", "conv5", "convt", "k_g4")): continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%d  mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
