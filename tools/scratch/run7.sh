#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/scratch/microbench_gconv.py > gpurun_out/mb_gconv7.txt 2>&1; grep -v amdgpu.ids gpurun_out/mb_gconv7.txt | grep "total\|B64.*L8192\|B64 1024\|B32 16->64 g4 L2049"
XFLAG="" KEEP_GOING=1 TAILN=30 bash tools/gpu_round.sh 2>&1 | grep -v "^    k_\|^    ms_" | cut -c1-2500
