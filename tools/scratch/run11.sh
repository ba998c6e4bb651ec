#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/music-synthesis_amd/featuresynth/_lib/variants
W="python3 tools/scratch/probe_atom_np.py --worker"
{ PROBE_DIL=3 timeout -k 10 120 $W; } > gpurun_out/probe11.txt 2>&1
grep "C=" gpurun_out/probe11.txt | cut -c1-150
timeout -k 10 300 python3 -m pytest tests/test_gpu_atom.py -q -p no:cacheprovider > gpurun_out/t11.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t11.txt
for i in 1 2; do
MSYNTH_LIB=$V/lib_rm0.so timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b11_old.json 2> gpurun_out/b11_old.log; echo "rm0: $(grep 'steps in' gpurun_out/b11_old.log)"
timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b11_new.json 2> gpurun_out/b11_new.log; echo "rm1: $(grep 'steps in' gpurun_out/b11_new.log)"
done
