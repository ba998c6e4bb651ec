#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_networks.py tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -2
A="--steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-gforward"
for i in 1 2 3; do
echo "== old"; MSYNTH_LIB=$R/music-synthesis_amd/featuresynth/_lib/libmsynth_hip.so python3 ab_old/bench.py $A 2>/dev/null | cut -c1-130
echo "== new"; python3 bench.py $A 2>/dev/null | cut -c1-130
echo "== new batch256"; DEBUG_HIP_GRAPH_BATCH_SIZE=256 python3 bench.py $A 2>/dev/null | cut -c1-130
echo "== new batch1024"; DEBUG_HIP_GRAPH_BATCH_SIZE=1024 python3 bench.py $A 2>/dev/null | cut -c1-130
echo "== new batch4"; DEBUG_HIP_GRAPH_BATCH_SIZE=4 python3 bench.py $A 2>/dev/null | cut -c1-130
done
