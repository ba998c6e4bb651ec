#!/bin/bash
# r04 GPU visit 1: atom scheme probe, atom tests, forked-replay graph topology + the forked replay test once
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 420 python3 tools/scratch/probe_atom_np.py > gpurun_out/probe_atom_np.txt 2>&1; echo "probe rc=$?"
timeout -k 10 300 python3 -m pytest tests/test_gpu_atom.py -q -p no:cacheprovider > gpurun_out/t_atom.txt 2>&1; echo "atom tests rc=$?"
tail -5 gpurun_out/t_atom.txt
timeout -k 10 240 python3 tools/dump_fork_graph.py gpurun_out/forkdot > gpurun_out/forkdot.txt 2>&1; echo "forkdot rc=$?"
MSYNTH_TEST_FORK_GRAPH=1 timeout -k 10 240 python3 -m pytest tests/test_gpu_realmelgan.py -q -p no:cacheprovider -k forked_replay > gpurun_out/t_fork.txt 2>&1; echo "fork test rc=$?"
tail -3 gpurun_out/t_fork.txt
