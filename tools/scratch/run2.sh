#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 0 1 2 3 4 5; do MSYNTH_ATOM_DBG=$d timeout -k 10 120 python3 tools/scratch/probe_atom.py; done > gpurun_out/probe_atom_dbg.txt 2>&1
cat gpurun_out/probe_atom_dbg.txt | grep MSYNTH
