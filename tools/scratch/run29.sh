#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
V=$R/music-synthesis_amd/featuresynth/_lib/variants/lib_addc.so
MSYNTH_LIB=$V timeout -k 10 600 python3 -m pytest tests/test_gpu_atom.py -x -q -m gpu -k "sign" 2>&1 | tail -3
A="--steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-gforward"
for i in 1 2 3; do
echo "== base"; python3 bench.py $A 2>/dev/null | cut -c1-130
echo "== addc"; MSYNTH_LIB=$V python3 bench.py $A 2>/dev/null | cut -c1-130
done
