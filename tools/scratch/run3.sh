#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/music-synthesis_amd/featuresynth/_lib/variants
W="python3 tools/scratch/probe_atom_np.py --worker"
{
for v in A B C D; do MSYNTH_LIB=$V/lib_$v.so timeout -k 10 120 $W; done
export PROBE_DIL=3
MSYNTH_LIB=$V/lib_D.so MSYNTH_ATOM_BALANCE=0 timeout -k 10 120 $W
for st in 4 8 16; do MSYNTH_LIB=$V/lib_D.so MSYNTH_ATOM_STAGGER=$st timeout -k 10 120 $W; done
for pr in 1 2; do MSYNTH_LIB=$V/lib_D.so MSYNTH_ATOM_PRIO=$pr timeout -k 10 120 $W; done
MSYNTH_LIB=$V/lib_A.so MSYNTH_ATOM_STAGGER=8 timeout -k 10 120 $W
MSYNTH_LIB=$V/lib_A.so MSYNTH_ATOM_PRIO=1 timeout -k 10 120 $W
} > gpurun_out/probe3.txt 2>&1
grep -c "us" gpurun_out/probe3.txt
