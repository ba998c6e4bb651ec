"""Weight gradient of the stride-8 transposed convs: split-bf16 kernel (wgrad_convt.hip) vs the fp32-MFMA phase-split
row kernel (MSYNTH_WGRADT8=0), both against float64."""
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
for (B, Cin, Lin, Cout, ia) in ((32, 512, 32, 256, 1), (32, 256, 256, 128, 1), (32, 256, 256, 128, 0), (3, 64, 64, 16, 1), (2, 128, 96, 48, 0)):
    K, S = 16, 8
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, K, device="cuda") * 0.02
    d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1, in_act=ia)
    gy = torch.randn(B, Cout, lo, device="cuda"); ya = torch.randn(B, Cout, lo, device="cuda")
    gp = torch.where(ya > 0, gy, gy * 0.2).double()
    xin = F.leaky_relu(x.double(), 0.2) if ia else x.double()
    wd = w.double().requires_grad_(True)
    F.conv_transpose1d(xin, wd, None, stride=S, padding=S // 2).backward(gp)
    ref, refb = wd.grad, gp.sum((0, 2))
    fl = 2.0 * B * Cin * Cout * K * Lin
    res = []
    for mode in ("0", "1"):
        os.environ["MSYNTH_WGRADT8"] = mode
        us, out = timeit(lambda: P.convt1d_bwd_weight(x, gy, ya, d, w.shape))
        gw, gb = out
        res.append((us, float((gw.double() - ref).norm() / ref.norm()), float((gb.double() - refb).norm() / refb.norm()),
                    L.load().ms_convt1d_kernel_name(d, 2).decode()))
    print("%-26s in_act %d fp32 %6.1f us %5.1f TF err %.1e/%.1e | split %6.1f us %5.1f TF x%.2f err %.1e/%.1e [%s]" % (
        (B, Cin, Lin, Cout), ia, res[0][0], fl / res[0][0] / 1e6, res[0][1], res[0][2], res[1][0], fl / res[1][0] / 1e6,
        res[0][0] / res[1][0], res[1][1], res[1][2], res[1][3]), flush=True)
    if B == 32 and ia: tot[0] += res[0][0]; tot[1] += res[1][0]
print("totals us (B = 32): fp32 %.0f split-bf16 %.0f" % tuple(tot))
