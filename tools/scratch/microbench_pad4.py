"""The D 1024->1024 k5 conv at the pooled scales (rows of 17 / 9 samples): rows padded to a multiple of 4 and run on
the aligned paired kernel (default) vs the dword loader of the four-wave kernel (MSYNTH_PAD4=0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    if isinstance(out, tuple): out = out[0]
    return e0.elapsed_time(e1) / n * 1e3, out

torch.manual_seed(0)
tot = [0.0, 0.0]
for (B, C, Lg) in ((64, 1024, 17), (32, 1024, 17), (64, 1024, 9), (32, 1024, 9), (64, 1024, 33), (7, 256, 21)):
    K = 5
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=2, act=1)
    fl = 2.0 * B * C * Lg * C * K
    gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda"); ga = torch.randn(B, C, Lg, device="cuda")
    for tag, fn, which in (("fwd", lambda: P.conv1d_fwd(x, w, b, d, lo), 0), ("bwd", lambda: P.conv1d_bwd_data(gy, ya, w, d, gx_add=ga), 1)):
        res = []
        for mode in ("0", "1"):
            os.environ["MSYNTH_PAD4"] = mode
            res.append(timeit(fn) + (L.load().ms_conv1d_kernel_name(d, which).decode(),))
        e = float((res[1][1] - res[0][1]).norm() / res[0][1].norm())
        print("%s %-16s unpadded %6.1f us %5.1f TF | padded %6.1f us %5.1f TF x%.2f diff %.1e [%s]" % (
            tag, (B, C, Lg), res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, res[0][0] / res[1][0], e, res[1][2]), flush=True)
        if C == 1024 and Lg < 30:
            tot[0] += res[0][0]; tot[1] += res[1][0]
print("totals us (L = 17 / 9): unpadded %.0f padded %.0f" % tuple(tot))
