#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/scratch/probe_stack.py 2>&1 | tail -8 && \
timeout -k 10 600 python3 -m pytest tests/test_gpu_stack.py -x -q -m gpu 2>&1 | tail -15
