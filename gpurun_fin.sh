mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_networks.py tests/test_gpu_dp.py tests/test_gpu_realmelgan.py tests/test_gpu_stage1.py -q -p no:cacheprovider 2>&1 | tail -2
bash tools/evidence_round.sh r05 a
