"""Grouped k41/s4 discriminator convs at the BASELINE shapes: time forward / backward-data / weight-grad
and report GB/s of algorithmic traffic.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

which = sys.argv[1:] or ["fwd", "bwd", "wgrad"]
B = int(os.environ.get("B", "32"))
tot = {k: 0.0 for k in which}
for L0 in (8192, 4097, 2049):
    Lin = L0
    for (Cin, Cout, groups) in ((16, 64, 4), (64, 256, 16), (256, 1024, 64), (1024, 1024, 256)):
        x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cout, 4, 41, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, stride=4, pad=20, groups=groups, act=1)
        y, _ = P.conv1d_fwd(x, w, b, d, lo)
        gy = torch.randn_like(y)
        nin, nout = x.numel() * 4, y.numel() * 4
        msg = "(%d,%d,%d,%d,g%d):" % (B, Cin, Lin, Cout, groups)
        if "fwd" in which:
            us = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo)); tot["fwd"] += us
            msg += "  fwd %6.1f us %5.0f GB/s" % (us, (nin + nout) / us / 1e3)
        if "bwd" in which:
            us = timeit(lambda: P.conv1d_bwd_data(gy, y, w, d)); tot["bwd"] += us
            msg += "  bwd %6.1f us %5.0f GB/s" % (us, (nin + 2 * nout) / us / 1e3)
        if "wgrad" in which:
            us = timeit(lambda: P.conv1d_bwd_weight(x, gy, y, d, w.shape)); tot["wgrad"] += us
            msg += "  wgrad %6.1f us %5.0f GB/s" % (us, (nin + 2 * nout) / us / 1e3)
        print(msg, flush=True)
        Lin = lo
print("totals (us):", {k: round(v, 1) for k, v in tot.items()})
