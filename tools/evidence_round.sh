#!/bin/bash
# One GPU-box visit that produces everything profiles/ holds for a round, on ONE box and ONE code version:
#   1. TCC traffic per kernel (tools/pmc_traffic.sh) -> profiles/<round>_pmc_traffic.json, in place BEFORE the bench so that the
#      bench line quotes it (same code_version)
#   2. serial + replay kernel stats and the bench line (tools/prof_final.sh)
#   3. MFMA / VALU / LDS counters of the dominant kernels (tools/pmc_kernels.sh)
#   4. the two-stage and RealMelGan bench lines
#   5. (part c) config 2: per-call listing + replay latency of the B = 1 generator forward and its kernels' counters
# usage (GPU box): bash tools/evidence_round.sh r05 [a|b|c|ab|abc]
set -o pipefail
tag=${1:-r05}
part=${2:-ab}          # a: traffic + traces + bench line; b: kernel counters + the two other bench lines (a call is at most 20 minutes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [[ $part == *a* ]]; then
bash tools/pmc_traffic.sh > gpurun_out/ev_pmc_traffic.log 2>&1; tail -3 gpurun_out/ev_pmc_traffic.log
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
bash tools/prof_final.sh > gpurun_out/ev_prof_final.log 2>&1; tail -4 gpurun_out/ev_prof_final.log
cut -c1-200 gpurun_out/final/bench.json
fi
if [[ $part == *b* ]]; then
bash tools/pmc_kernels.sh > gpurun_out/ev_pmc_kernels.log 2>&1; tail -3 gpurun_out/ev_pmc_kernels.log
timeout -k 10 300 python3 bench.py --model twostage --steps 20 --warmup 6 > gpurun_out/ev_bench_twostage.json 2> gpurun_out/ev_bench_twostage.log; echo "twostage rc=$?"
timeout -k 10 300 python3 bench.py --model realmelgan --steps 20 --warmup 6 > gpurun_out/ev_bench_realmelgan.json 2> gpurun_out/ev_bench_realmelgan.log; echo "realmelgan rc=$?"
cut -c1-200 gpurun_out/ev_bench_twostage.json; cut -c1-200 gpurun_out/ev_bench_realmelgan.json
fi
if [[ $part == *c* ]]; then
timeout -k 10 200 python3 tools/gfwd_b1.py --list > gpurun_out/${tag}_gfwd_b1_list.txt 2> gpurun_out/ev_gfwd_b1.log; echo "gfwd_b1 rc=$?"
bash tools/pmc_gfwd_b1.sh $tag > gpurun_out/ev_pmc_gfwd_b1.log 2>&1; tail -3 gpurun_out/ev_pmc_gfwd_b1.log
cp gpurun_out/${tag}_gfwd_b1_list.txt gpurun_out/${tag}_pmc_gfwd_b1.txt profiles/ 2>/dev/null
fi
