"""One grouped-conv shape in a loop, for rocprofv3 --pmc runs: python3 tools/pmc_one.py fwd|bwd|wgrad B Cin Lin Cout groups"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P
op = sys.argv[1]
if op == "dwgrad":      # dense weight gradient: dwgrad B Cin L Cout K dil
    B, Cin, Lg, Cout, K, dil = map(int, sys.argv[2:8])
    x = torch.randn(B, Cin, Lg, device="cuda"); gy = torch.randn(B, Cout, Lg, device="cuda"); ya = torch.randn(B, Cout, Lg, device="cuda")
    d, lo = P.conv_desc(x.shape, (Cout, Cin, K), pad=dil * (K - 1) // 2, dil=dil, act=1)
    for _ in range(5): P.conv1d_bwd_weight(x, gy, ya, d, (Cout, Cin, K))
    torch.cuda.synchronize(); sys.exit(0)
if op == "dfwd":      # dense forward: dfwd B C L K dil
    B, C, Lg, K, dil = map(int, sys.argv[2:7])
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    for _ in range(5): P.conv1d_fwd(x, w, b, d, lo)
    torch.cuda.synchronize(); sys.exit(0)
B, Cin, Lin, Cout, groups = map(int, sys.argv[2:7])
x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cout, 4, 41, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
d, lo = P.conv_desc(x.shape, w.shape, stride=4, pad=20, groups=groups, act=1)
y, _ = P.conv1d_fwd(x, w, b, d, lo); gy = torch.randn_like(y)
for _ in range(5):
    if op == "fwd": P.conv1d_fwd(x, w, b, d, lo)
    elif op == "bwd": P.conv1d_bwd_data(gy, y, w, d)
    else: P.conv1d_bwd_weight(x, gy, y, d, w.shape)
torch.cuda.synchronize()
