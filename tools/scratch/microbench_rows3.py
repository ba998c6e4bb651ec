"""Dense row-tile convs at the BASELINE shapes: split-bf16 kernel (conv_rows3.hip) vs the fp32-MFMA pipelined
kernel (MSYNTH_ROWS3=0), with the difference between the two."""
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

def both(tag, fl, fn, d, which):
    res = []
    for mode in ("0", "1", "2"):
        os.environ["MSYNTH_ROWS3"] = "0" if mode == "0" else "1"
        os.environ["MSYNTH_ROWS3P"] = "1" if mode == "2" else "0"
        name = L.load().ms_conv1d_kernel_name(d, which).decode()
        res.append(timeit(fn) + (name,))
    e = float((res[2][1] - res[0][1]).norm() / res[0][1].norm())
    print("%-34s fp32 %6.1f us %5.1f TF | split4w %6.1f us %5.1f TF x%.2f | paired %6.1f us %5.1f TF x%.2f  diff %.1e [%s]" % (
        tag, res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, res[0][0] / res[1][0],
        res[2][0], fl / res[2][0] / 1e6, res[0][0] / res[2][0], e, res[2][2]), flush=True)
    return res[0][0], min(res[1][0], res[2][0])

torch.manual_seed(0)
tot = [0.0, 0.0]
for (B, C, Lg, K, dil) in ((32, 128, 2048, 3, 1), (32, 128, 2048, 3, 9), (32, 256, 256, 3, 1), (32, 256, 256, 3, 9), (32, 64, 4096, 3, 3),
                           (32, 32, 8192, 3, 1), (64, 1024, 32, 5, 1), (32, 1024, 32, 5, 1), (64, 1024, 16, 5, 1)):
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    res = torch.randn(B, C, Lg, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    fl = 2.0 * B * C * Lg * C * K
    a = both("fwd %s" % ((B, C, Lg, K, dil),), fl, lambda: P.conv1d_fwd(x, w, b, d, lo, residual=res, want_y_act=True), d, 0)
    gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
    c = both("bwd_data %s" % ((B, C, Lg, K, dil),), fl, lambda: P.conv1d_bwd_data(gy, ya, w, d, gx_add=res), d, 1)
    tot[0] += a[0] + c[0]; tot[1] += a[1] + c[1]
print("totals us: fp32 %.0f split-bf16 %.0f" % tuple(tot))
print("-- the k5 conv at the pooled scales (rows of 17 / 9 samples)")
for (B, C, Lg, K, dil) in ((64, 1024, 17, 5, 1), (32, 1024, 17, 5, 1), (64, 1024, 9, 5, 1), (32, 1024, 9, 5, 1)):
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    fl = 2.0 * B * C * Lg * C * K
    both("fwd %s" % ((B, C, Lg, K, dil),), fl, lambda: P.conv1d_fwd(x, w, b, d, lo), d, 0)
    gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
    both("bwd_data %s" % ((B, C, Lg, K, dil),), fl, lambda: P.conv1d_bwd_data(gy, ya, w, d), d, 1)
