"""Dense weight gradients at the BASELINE shapes: row-tile kernel vs the im2col kernel (MSYNTH_WROWS=0)."""
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
    return e0.elapsed_time(e1) / n * 1e3, out

torch.manual_seed(0)
shapes = [(32, 32, 8192, 32, 3, 1), (32, 32, 8192, 32, 3, 9), (32, 128, 2048, 128, 3, 1), (32, 128, 2048, 128, 3, 9), (32, 256, 256, 256, 3, 1), (32, 256, 256, 256, 3, 9),
          (32, 64, 4096, 64, 3, 1), (32, 64, 4096, 64, 3, 3), (64, 1024, 32, 1024, 5, 1), (64, 1024, 17, 1024, 5, 1),
          (64, 1024, 9, 1024, 5, 1), (32, 512, 32, 512, 3, 1)]
tot = [0.0, 0.0]
for (B, Cin, Lg, Cout, K, dil) in shapes:
    x = torch.randn(B, Cin, Lg, device="cuda"); gy = torch.randn(B, Cout, Lg, device="cuda"); ya = torch.randn(B, Cout, Lg, device="cuda")
    d, lo = P.conv_desc(x.shape, (Cout, Cin, K), pad=dil * (K - 1) // 2, dil=dil, act=1)
    fl = 2.0 * B * Cout * Lg * Cin * K
    res = []
    for mode in ("0", None):
        if mode is None: os.environ.pop("MSYNTH_WROWS", None)
        else: os.environ["MSYNTH_WROWS"] = mode
        us, out = timeit(lambda: P.conv1d_bwd_weight(x, gy, ya, d, (Cout, Cin, K)))
        name = L.load().ms_conv1d_kernel_name(d, 2).decode()
        res.append((us, out, name))
    tot[0] += res[0][0]; tot[1] += res[1][0]
    gw0, gb0 = res[0][1][0], res[0][1][1]; gw1, gb1 = res[1][1][0], res[1][1][1]
    e = float((gw1 - gw0).norm() / gw0.norm()); eb = float((gb1 - gb0).norm() / gb0.norm())
    print("%-28s im2col %7.1f us %5.1f TF/s | rows %7.1f us %5.1f TF/s  (%s)  rel diff %.1e / %.1e" % (
        (B, Cin, Lg, Cout, K, dil), res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, res[1][2], e, eb), flush=True)
print("totals us: im2col %.0f rows %.0f" % tuple(tot))
