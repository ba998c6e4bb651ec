#!/bin/bash
# The oracle's C restatement (oracle/msynth_oracle.c) under AddressSanitizer + UBSan on the CPU: builds an instrumented
# libmsynth_oracle.so in place, runs tests/test_oracle_golden.py against the committed golden vectors, restores the normal build.
# (GPU sanitizers are not available on the pool; the HIP side has no CPU build.)   usage (here): bash tools/oracle_sanitize.sh
set -o pipefail
cd "$(dirname "$0")/.."
keep=$(mktemp) && cp oracle/libmsynth_oracle.so "$keep" 2>/dev/null
trap 'if [ -s "$keep" ]; then cp "$keep" oracle/libmsynth_oracle.so; else make -s -B -C oracle; fi; touch oracle/libmsynth_oracle.so; rm -f "$keep"' EXIT
gcc -O1 -g -fPIC -fopenmp -std=c11 -fno-fast-math -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o oracle/libmsynth_oracle.so oracle/msynth_oracle.c -lm || exit 1
touch oracle/libmsynth_oracle.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    timeout 1200 python -m pytest tests/test_oracle_golden.py -x -q
