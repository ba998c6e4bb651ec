"""Dense k3 weight gradients at the BASELINE shapes: split-bf16 kernel (k_wgrad_rows3) vs the fp32-MFMA row kernel
(MSYNTH_WROWS3=0): single layers and the batched six-layer launch of a ResidualStack."""
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
for (B, C, Lg) in ((32, 128, 2048), (32, 256, 256), (32, 64, 4096)):
    x = [torch.randn(B, C, Lg, device="cuda") for _ in range(6)]
    gy = [torch.randn(B, C, Lg, device="cuda") for _ in range(6)]
    ya = [torch.randn(B, C, Lg, device="cuda") for _ in range(6)]
    dils = (1, 9, 1, 3, 1, 1)
    ds = [P.conv_desc(x[0].shape, (C, C, 3), pad=d, dil=d, act=1)[0] for d in dils]
    fl1 = 2.0 * B * C * Lg * C * 3
    res = {}
    for mode in ("0", "1"):
        os.environ["MSYNTH_WROWS3"] = mode
        name = L.load().ms_conv1d_kernel_name(ds[1], 2).decode()
        us1, o1 = timeit(lambda: P.conv1d_bwd_weight(x[1], gy[1], ya[1], ds[1], (C, C, 3)))
        jobs = [(x[i], gy[i], ya[i], ds[i], (C, C, 3), None, None, False) for i in range(6)]
        us6, o6 = timeit(lambda: P.conv1d_bwd_weight_multi(jobs))
        res[mode] = (us1, o1, us6, o6, name)
    a, b = res["0"], res["1"]
    e1 = float((b[1][0] - a[1][0]).norm() / a[1][0].norm()); eb = float((b[1][1] - a[1][1]).norm() / a[1][1].norm())
    e6 = max(float((b[3][i][0] - a[3][i][0]).norm() / a[3][i][0].norm()) for i in range(6))
    eb6 = max(float((b[3][i][1] - a[3][i][1]).norm() / a[3][i][1].norm()) for i in range(6))
    print("%-16s single d9: fp32 %6.1f us %5.1f TF | split %6.1f us %5.1f TF x%.2f (diff %.1e / bias %.1e)  || stack of 6: fp32 %6.1f us %5.1f TF | split %6.1f us %5.1f TF x%.2f (diff %.1e / %.1e) [%s]" % (
        (B, C, Lg), a[0], fl1 / a[0] / 1e6, b[0], fl1 / b[0] / 1e6, a[0] / b[0], e1, eb,
        a[2], 6 * fl1 / a[2] / 1e6, b[2], 6 * fl1 / b[2] / 1e6, a[2] / b[2], e6, eb6, b[4]), flush=True)
