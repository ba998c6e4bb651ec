#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/music-synthesis_amd/featuresynth/_lib/variants
for i in 1 2 3; do
MSYNTH_LIB=$V/lib_gold.so timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/bench8_old.json 2> gpurun_out/bench8_old.log; echo "gconv bf16x3: $(grep 'steps in' gpurun_out/bench8_old.log)"
timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/bench8_new.json 2> gpurun_out/bench8_new.log; echo "gconv fp16x2: $(grep 'steps in' gpurun_out/bench8_new.log)"
done
timeout -k 10 120 python3 tools/gfwd_b1.py --list > gpurun_out/gfwd_b1.txt 2>&1; grep -v amdgpu gpurun_out/gfwd_b1.txt
