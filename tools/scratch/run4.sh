#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/music-synthesis_amd/featuresynth/_lib/variants
W="python3 tools/scratch/probe_atom_np.py --worker"
{ PROBE_C=32 timeout -k 10 120 $W; for v in E1 E3; do PROBE_C=32 MSYNTH_LIB=$V/lib_$v.so timeout -k 10 120 $W; done; } > gpurun_out/probe4.txt 2>&1
grep "C=32" gpurun_out/probe4.txt | cut -c1-140
XFLAG="" KEEP_GOING=1 TAILN=60 bash tools/gpu_round.sh
