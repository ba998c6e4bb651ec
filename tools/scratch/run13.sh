#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_stage1.py -q -p no:cacheprovider -k two_stage -s > gpurun_out/t13.txt 2>&1; echo "two-stage test rc=$?"; grep -v amdgpu gpurun_out/t13.txt | tail -15
W="python3 tools/scratch/probe_atom_np.py --worker"
{ PROBE_DIL=9 PROBE_C=128 timeout -k 10 120 $W; PROBE_DIL=9 PROBE_C=128 MSYNTH_ATOM_BWD9=1 timeout -k 10 120 $W; PROBE_DIL=9 PROBE_C=256 timeout -k 10 120 $W; PROBE_DIL=9 PROBE_C=256 MSYNTH_ATOM_BWD9=1 timeout -k 10 120 $W; } > gpurun_out/probe13.txt 2>&1
grep "C=" gpurun_out/probe13.txt | cut -c1-160
python3 - <<'PY'
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "music-synthesis_amd")
import torch
from featuresynth._ops import graph as G, prims as P
# two-launch backward at d = 9 for comparison
for C, Lg in ((128, 2048), (256, 256)):
    B, dil = 32, 9
    x = torch.randn(B, C, Lg, device="cuda"); w0 = torch.randn(C, C, 3, device="cuda") * 0.05; w1 = torch.randn(C, C, 3, device="cuda") * 0.05
    b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
    y, rec = G.atom_forward(x, w0, b0, w1, b1, dil, True, image=None)
    d0, d1, _, t, u = rec
    g = torch.randn_like(x) * 1e-6
    def two():
        gt = P.conv1d_bwd_data(g, u, w1, d1)
        return P.conv1d_bwd_data(gt, t, w0, d0, gx_add=g)
    for _ in range(4): two()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(30): two()
    e1.record(); torch.cuda.synchronize()
    print("C=%d d=9 two-launch backward: %.1f us" % (C, e0.elapsed_time(e1) / 30 * 1e3))
PY
