#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/music-synthesis_amd/featuresynth/_lib/variants
W="python3 tools/scratch/probe_atom_np.py --worker"
{ for l in head pre0; do PROBE_DIL=3 MSYNTH_LIB=$V/lib_$l.so timeout -k 10 120 $W; done; PROBE_DIL=3 timeout -k 10 120 $W; } > gpurun_out/probe14.txt 2>&1
grep "C=" gpurun_out/probe14.txt | cut -c1-150
timeout -k 10 500 python3 -m pytest tests/test_gpu_atom.py tests/test_gpu_conv5.py tests/test_gpu_ops.py -q -p no:cacheprovider > gpurun_out/t14.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t14.txt
for i in 1 2; do
for l in head pre0; do MSYNTH_LIB=$V/lib_$l.so timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b14_$l.json 2> gpurun_out/b14_$l.log; echo "$l: $(grep 'steps in' gpurun_out/b14_$l.log)"; done
timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b14_new.json 2> gpurun_out/b14_new.log; echo "new: $(grep 'steps in' gpurun_out/b14_new.log)"
done
