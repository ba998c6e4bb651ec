"""Timing probes of the paired split-bf16 conv kernel (k_conv_rows3p<2,2,2,3,0>, C = 128 atoms): MSYNTH_R3P_DBG =
0 product kernel | 1 staging without the operand split | 2 no MFMAs | 3 no staging (MFMAs + fragment reads only).
One process per variant (the switch is read once):  for d in 0 1 2 3; do MSYNTH_R3P_DBG=$d python3 tools/probe_r3p.py; done"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P

B, C, Lg, K, dil = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 128, 2048, 3, 1))]
x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
for _ in range(5):
    P.conv1d_fwd(x, w, b, d, lo)
torch.cuda.synchronize()
n = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    P.conv1d_fwd(x, w, b, d, lo)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * C * Lg * C * K
print("MSYNTH_R3P_DBG=%s %s: %.1f us/launch (back to back), %.1f TFLOP/s nominal" % (
    os.environ.get("MSYNTH_R3P_DBG", "0"), (B, C, Lg, K, dil), us, fl / us / 1e6), flush=True)
