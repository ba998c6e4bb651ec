#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== 2-rank rehearsal on one GPU (gloo, shared card: control flow only)"
MSYNTH_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -4 | cut -c1-400
echo "== pmc kernels"
bash tools/pmc_kernels.sh > gpurun_out/ev_pmc_kernels.log 2>&1; grep -A1 "k_atom_fwd" gpurun_out/r04_pmc_kernels_summary.txt | cut -c1-330
