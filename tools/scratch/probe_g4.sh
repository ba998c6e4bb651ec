#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/scratch/probe_determinism.py > gpurun_out/pg4_new.log 2>&1; grep "^run\|differing" gpurun_out/pg4_new.log | cut -c1-160
timeout -k 10 200 python tools/scratch/probe_determinism.py > gpurun_out/pg4_new2.log 2>&1; grep "^run\|differing" gpurun_out/pg4_new2.log | cut -c1-160
timeout -k 10 600 python -m pytest tests/test_gpu_parts.py tests/test_gpu_dp.py -q -x > gpurun_out/t_g4.log 2>&1; tail -4 gpurun_out/t_g4.log
