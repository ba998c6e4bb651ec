#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "grouped or conv" > gpurun_out/t_ops.txt 2>&1; echo "ops tests rc=$?"; tail -8 gpurun_out/t_ops.txt
timeout -k 10 200 python3 tools/scratch/microbench_gconv.py fwd > gpurun_out/mb_gconv.txt 2>&1; grep -v amdgpu.ids gpurun_out/mb_gconv.txt | tail -30
for i in 1 2; do
for np in 3 2; do MSYNTH_C5_NP=$np timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/bench6_$np.json 2> gpurun_out/bench6_$np.log; echo "C5_NP=$np: $(grep 'steps in' gpurun_out/bench6_$np.log)"; done
done
