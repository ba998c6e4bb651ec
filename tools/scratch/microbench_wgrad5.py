"""Weight gradient of the many-channel k5 conv on short rows: split-bf16 kernel (wgrad_k5.hip) vs the fp32-MFMA row-tile
kernel (MSYNTH_WGRAD5=0), both against float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
import torch.nn.functional as F
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out

torch.manual_seed(0)
tot = [0.0, 0.0]
for (B, Cin, Cout, Lg) in ((64, 1024, 1024, 32), (64, 1024, 1024, 17), (64, 1024, 1024, 9), (3, 256, 320, 33), (5, 320, 256, 12), (2, 256, 256, 7)):
    x = torch.randn(B, Cin, Lg, device="cuda"); w = torch.randn(Cout, Cin, 5, device="cuda") * 0.02
    d, lo = P.conv_desc(x.shape, w.shape, pad=2, act=1)
    gy = torch.randn(B, Cout, Lg, device="cuda"); ya = torch.randn(B, Cout, Lg, device="cuda")
    gp = torch.where(ya > 0, gy, gy * 0.2).double()
    wd = w.double().requires_grad_(True)
    F.conv1d(x.double(), wd, None, padding=2).backward(gp)
    ref, refb = wd.grad, gp.sum((0, 2))
    fl = 2.0 * B * Cin * Cout * 5 * Lg
    res = []
    for mode in ("0", "1"):
        os.environ["MSYNTH_WGRAD5"] = mode
        us, out = timeit(lambda: P.conv1d_bwd_weight(x, gy, ya, d, w.shape))
        gw, gb = out
        res.append((us, float((gw.double() - ref).norm() / ref.norm()), float((gb.double() - refb).norm() / refb.norm()),
                    L.load().ms_conv1d_kernel_name(d, 2).decode()))
    print("%-22s fp32 %6.1f us %5.1f TF err %.1e/%.1e | split %6.1f us %5.1f TF x%.2f err %.1e/%.1e [%s]" % (
        (B, Cin, Cout, Lg), res[0][0], fl / res[0][0] / 1e6, res[0][1], res[0][2], res[1][0], fl / res[1][0] / 1e6,
        res[0][0] / res[1][0], res[1][1], res[1][2], res[1][3]), flush=True)
    if B == 64: tot[0] += res[0][0]; tot[1] += res[1][0]
print("totals us (B = 64): fp32 %.0f split-bf16 %.0f" % tuple(tot))
