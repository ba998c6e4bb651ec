#!/bin/bash
for gw in 2048 4096 6144 8192 16384; do echo "== B=128 MSYNTH_GW=$gw"; B=128 MSYNTH_GW=$gw timeout -k 10 120 python tools/microbench_gconv.py fwd 2>&1 | grep -E "^\(|totals"; done
