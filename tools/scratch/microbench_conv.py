import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L
B, Lg, K, dil = 32, 2048, 3, 3
for Cout in (128, 64):
  for Cin in (32, 64, 128, 256, 512):
    x = torch.randn(B, Cin, Lg, device="cuda"); w = torch.randn(Cout, Cin, K, device="cuda") * 0.05; b = torch.zeros(Cout, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil, dil=dil, act=1)
    for _ in range(3): P.conv1d_fwd(x, w, b, d, lo)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(20): P.conv1d_fwd(x, w, b, d, lo)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * B * Cout * Lg * Cin * K
    print("Cout=%d Cin=%d: %.1f us  %.1f TF/s  (%s)" % (Cout, Cin, ms * 1e3, fl / ms / 1e9, L.load().ms_conv1d_kernel_name(d, 0).decode()))
