#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv5.py -q -p no:cacheprovider > gpurun_out/t_conv5.txt 2>&1; echo "conv5 tests rc=$?"; tail -15 gpurun_out/t_conv5.txt
for np in 3 2; do MSYNTH_C5_NP=$np timeout -k 10 200 python3 tools/scratch/microbench_conv5.py; done > gpurun_out/mb_conv5.txt 2>&1; cat gpurun_out/mb_conv5.txt | grep -v amdgpu.ids
timeout -k 10 300 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/bench5.json 2> gpurun_out/bench5.log; grep "steps in" gpurun_out/bench5.log
