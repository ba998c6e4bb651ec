"""k_gconv_split_fwd on the long layers under the launch-shape knob MSYNTH_GW (target waves per launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, Cin, Cout, groups, Lin) in ((64, 16, 64, 4, 8192), (32, 16, 64, 4, 8192), (64, 256, 1024, 64, 512), (64, 64, 256, 16, 1025), (128, 16, 64, 4, 8192)):
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cout, 4, 41, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, stride=4, pad=20, groups=groups, act=1)
    row = []
    for gw in ("512", "1024", "1536", "2048", "3072", "4096"):
        os.environ["MSYNTH_GW"] = gw
        row.append("%s:%5.1f" % (gw, timeit(lambda: P.conv1d_fwd(x, w, b, d, lo))))
    print((B, Cin, Lin), " ".join(row), flush=True)
