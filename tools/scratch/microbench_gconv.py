"""Grouped k41 / stride-4 convs of the discriminator at the BASELINE shapes (3 scales x 4 layers, B = 64 and 32):
split-bf16 kernels (gconv_split.hip) vs the fp32-MFMA kernels (MSYNTH_GCONV3=0), each checked against float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
import torch.nn.functional as F
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    if isinstance(out, tuple): out = out[0]
    return e0.elapsed_time(e1) / n * 1e3, out

def rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm())

def both(tag, fl, nbytes, fn, d, which, ref):
    res = []
    for mode in ("0", "1"):
        os.environ["MSYNTH_GCONV3"] = mode
        name = L.load().ms_conv1d_kernel_name(d, which).decode()
        res.append(timeit(fn) + (name,))
    print("%-40s fp32 %6.1f us %5.1f TF err %.1e | split %6.1f us %5.1f TF %4.2f TB/s x%.2f err %.1e [%s]" % (
        tag, res[0][0], fl / res[0][0] / 1e6, rel(res[0][1], ref), res[1][0], fl / res[1][0] / 1e6,
        nbytes / res[1][0] / 1e6, res[0][0] / res[1][0], rel(res[1][1], ref), res[1][2]), flush=True)
    return res[0][0], res[1][0]

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fwd", "bwd", "wgrad"]
torch.manual_seed(0)
tot = {k: [0.0, 0.0] for k in which}
LAYERS = ((16, 64, 4), (64, 256, 16), (256, 1024, 64), (1024, 1024, 256))
for B in (64, 32):
    for L0 in (8192, 4097, 2049):
        Lin = L0
        for (Cin, Cout, groups) in LAYERS:
            x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cout, 4, 41, device="cuda") * 0.05
            b = torch.randn(Cout, device="cuda")
            d, lo = P.conv_desc(x.shape, w.shape, stride=4, pad=20, groups=groups, act=1)
            fl = 2.0 * B * Cout * lo * 4 * 41
            nb = 4.0 * (x.numel() + B * Cout * lo)
            small = B * Cin * Lin <= (1 << 23)
            pre64 = F.conv1d(x.double(), w.double(), b.double(), stride=4, padding=20, groups=groups) if small else None
            tag = "B%d %d->%d g%d L%d->%d" % (B, Cin, Cout, groups, Lin, lo)
            if "fwd" in which:
                y64 = F.leaky_relu(pre64, 0.2) if small else P.conv1d_fwd(x, w, b, d, lo).double()
                r = both("fwd " + tag, fl, nb, lambda: P.conv1d_fwd(x, w, b, d, lo), d, 0, y64)
                tot["fwd"][0] += r[0]; tot["fwd"][1] += r[1]
            gy = torch.randn(B, Cout, lo, device="cuda"); ya = torch.randn(B, Cout, lo, device="cuda")
            if "bwd" in which or "wgrad" in which:
                gp = torch.where(ya > 0, gy, gy * 0.2).double()
            if "bwd" in which:
                if small:
                    ref = F.conv_transpose1d(gp, w.double(), stride=4, padding=20, groups=groups,
                                             output_padding=Lin - ((lo - 1) * 4 - 40 + 41))
                else:
                    os.environ["MSYNTH_GCONV3"] = "0"; ref = P.conv1d_bwd_data(gy, ya, w, d).double()
                r = both("bwd " + tag, fl, nb + 4.0 * gy.numel(), lambda: P.conv1d_bwd_data(gy, ya, w, d), d, 1, ref)
                tot["bwd"][0] += r[0]; tot["bwd"][1] += r[1]
            if "wgrad" in which:
                if small:
                    xd = x.double().requires_grad_(False); wd = w.double().requires_grad_(True)
                    F.conv1d(xd, wd, None, stride=4, padding=20, groups=groups).backward(gp)
                    ref = wd.grad
                else:
                    os.environ["MSYNTH_GCONV3"] = "0"; ref = P.conv1d_bwd_weight(x, gy, ya, d, w.shape)[0].double()
                r = both("wgrad " + tag, fl, nb + 4.0 * gy.numel(), lambda: P.conv1d_bwd_weight(x, gy, ya, d, w.shape), d, 2, ref)
                tot["wgrad"][0] += r[0]; tot["wgrad"][1] += r[1]
            Lin = lo
for k in which:
    print("total %s us: fp32 %.0f split-bf16 %.0f" % (k, tot[k][0], tot[k][1]))
