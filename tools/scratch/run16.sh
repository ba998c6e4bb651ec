#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/music-synthesis_amd/featuresynth/_lib/variants
{ echo "== old (bf16 x 3)"; MSYNTH_CONVTIMG_S2=1 MSYNTH_LIB=$V/lib_ctold.so timeout -k 10 120 python3 tools/scratch/microbench_convt_img.py; echo "== new (fp16 x 2)"; MSYNTH_CONVTIMG_S2=1 timeout -k 10 120 python3 tools/scratch/microbench_convt_img.py; } > gpurun_out/mb16.txt 2>&1; grep -v amdgpu gpurun_out/mb16.txt | grep "==\|B=32"
timeout -k 10 300 python3 -m pytest tests/test_gpu_convt_img.py tests/test_gpu_stage1.py -q -p no:cacheprovider > gpurun_out/t16.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t16.txt
for i in 1 2 3; do
MSYNTH_LIB=$V/lib_ctold.so timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b16_old.json 2> gpurun_out/b16_old.log; echo "convt bf16x3: $(grep 'steps in' gpurun_out/b16_old.log)"
timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b16_new.json 2> gpurun_out/b16_new.log; echo "convt fp16x2: $(grep 'steps in' gpurun_out/b16_new.log)"
done
MSYNTH_CONVTIMG_S2=1 timeout -k 10 300 python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline > gpurun_out/b16_s2.json 2> gpurun_out/b16_s2.log; echo "convt fp16x2 + stride 2 on the image kernel: $(grep 'steps in' gpurun_out/b16_s2.log)"
