#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/scratch/microbench_wmulti_np.py > gpurun_out/mb_w15.txt 2>&1; grep -v amdgpu gpurun_out/mb_w15.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_atom.py tests/test_gpu_networks.py tests/test_gpu_ops.py -q -p no:cacheprovider -x > gpurun_out/t15.txt 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/t15.txt
for i in 1 2 3; do
MSYNTH_WROWS3_NP=3 timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b15_old.json 2> gpurun_out/b15_old.log; echo "wgrad bf16x3: $(grep 'steps in' gpurun_out/b15_old.log)"
timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b15_new.json 2> gpurun_out/b15_new.log; echo "wgrad fp16x2: $(grep 'steps in' gpurun_out/b15_new.log)"
done
