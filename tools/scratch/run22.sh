#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_stack.py tests/test_gpu_networks.py -x -q -m gpu 2>&1 | tail -5 || exit 1
A="--steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-gforward"
for i in 1 2 3; do
echo "== per atom"; MSYNTH_STACK=0 python3 bench.py $A 2>/dev/null | cut -c1-130
echo "== stacks"; python3 bench.py $A 2>/dev/null | cut -c1-130
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('generator_forward_b1'), indent=1))"
