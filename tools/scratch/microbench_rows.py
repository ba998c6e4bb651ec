"""Row-tile forward / backward-data kernels at the BASELINE shapes: pipelined kernel vs the first generation
(MSYNTH_ROWS2=0), with the max difference between the two."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    if isinstance(out, tuple): out = out[0]
    return e0.elapsed_time(e1) / n * 1e3, out

def both(tag, fl, fn):
    res = []
    for mode in ("0", None):
        if mode is None: os.environ.pop("MSYNTH_ROWS2", None)
        else: os.environ["MSYNTH_ROWS2"] = mode
        res.append(timeit(fn))
    e = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print("%-40s gen1 %7.1f us %5.1f TF/s | gen2 %7.1f us %5.1f TF/s   rel maxdiff %.1e" % (
        tag, res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, e), flush=True)
    return res[0][0], res[1][0]

torch.manual_seed(0)
tot = [0.0, 0.0]
for (B, C, Lg, K, dil) in ((32, 128, 2048, 3, 1), (32, 128, 2048, 3, 9), (32, 256, 256, 3, 1), (32, 256, 256, 3, 9), (32, 64, 4096, 3, 3),
                           (32, 32, 8192, 3, 1), (64, 1024, 32, 5, 1), (32, 1024, 32, 5, 1), (32, 512, 32, 3, 1)):
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    res = torch.randn(B, C, Lg, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    fl = 2.0 * B * C * Lg * C * K
    a = both("fwd %s" % ((B, C, Lg, K, dil),), fl, lambda: P.conv1d_fwd(x, w, b, d, lo, residual=res, want_y_act=True))
    gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
    c = both("bwd_data %s" % ((B, C, Lg, K, dil),), fl, lambda: P.conv1d_bwd_data(gy, ya, w, d, gx_add=res))
    tot[0] += a[0] + c[0]; tot[1] += a[1] + c[1]
for (Cin, Lin, Cout, K, S) in ((512, 32, 256, 16, 8), (256, 256, 128, 16, 8), (128, 2048, 64, 4, 2), (64, 4096, 32, 4, 2)):
    B = 32
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, K, device="cuda") * 0.02; b = torch.randn(Cout, device="cuda")
    d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1)
    fl = 2.0 * B * Cin * Cout * K * Lin
    a = both("convT fwd %s" % ((Cin, Lin, Cout, K, S),), fl, lambda: P.convt1d_fwd(x, w, b, d, lo))
    tot[0] += a[0]; tot[1] += a[1]
print("totals us: gen1 %.0f gen2 %.0f" % tuple(tot))
