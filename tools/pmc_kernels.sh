#!/bin/bash
# MFMA-busy / VALU / LDS-conflict counters of the dominant kernels at their bench shapes:
# the fused atom (C = 64 and 128), the k5 layer on weight images (B = 64, L = 32), transposed-conv backward data (256 <- 128, stride 8) the grouped k41 convs and the 256-group layer on v_mfma_f32_4x4x1 (gconv4.hip).
# Separate --pmc passes, kernel-trace only (tools/pmc_one.sh).  Output: gpurun_out/pmc_<tag>.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_one.sh atom64 atom 32 64 4096 3 > gpurun_out/pmc_atom64.txt 2>&1
bash tools/pmc_one.sh atom128 atom 32 128 2048 1 > gpurun_out/pmc_atom128.txt 2>&1
bash tools/pmc_one.sh k5 k5parts 64 > gpurun_out/pmc_k5.txt 2>&1
bash tools/pmc_one.sh ctbwd ctbwd 32 256 256 128 8 > gpurun_out/pmc_ctbwd.txt 2>&1
bash tools/pmc_one.sh gfwd fwd 64 64 2048 256 16 > gpurun_out/pmc_gfwd.txt 2>&1
bash tools/pmc_one.sh gwgrad wgrad 64 64 2048 256 16 > gpurun_out/pmc_gwgrad.txt 2>&1
bash tools/pmc_one.sh g4 g4 64 > gpurun_out/pmc_g4.txt 2>&1
bash tools/pmc_one.sh gen gen 32 > gpurun_out/pmc_gen.txt 2>&1      # every generator kernel of a training pass (atoms at all four widths, batched weight gradients)
python3 tools/pmc_ratios.py gpurun_out/pmc_atom64.txt gpurun_out/pmc_atom128.txt gpurun_out/pmc_k5.txt gpurun_out/pmc_ctbwd.txt gpurun_out/pmc_gfwd.txt gpurun_out/pmc_gwgrad.txt gpurun_out/pmc_g4.txt gpurun_out/pmc_gen.txt > gpurun_out/r05_pmc_kernels_summary.txt
cat gpurun_out/r05_pmc_kernels_summary.txt
