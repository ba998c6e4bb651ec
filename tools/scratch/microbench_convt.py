"""ConvTranspose1d weight gradient: row-tile phase-split kernel vs the im2col kernel (MSYNTH_WROWS=0)."""
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
for (Cin, Lin, Cout, K, S) in ((512, 32, 256, 16, 8), (256, 256, 128, 16, 8), (128, 2048, 64, 4, 2), (64, 4096, 32, 4, 2)):
    B = 32
    x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, K, device="cuda") * 0.02
    d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1)
    gy = torch.randn(B, Cout, lo, device="cuda"); ya = torch.randn(B, Cout, lo, device="cuda")
    fl = 2.0 * B * Cin * Cout * K * Lin
    res = []
    for mode in ("0", None):
        if mode is None: os.environ.pop("MSYNTH_WROWS", None)
        else: os.environ["MSYNTH_WROWS"] = mode
        us, out = timeit(lambda: P.convt1d_bwd_weight(x, gy, ya, d, w.shape))
        res.append((us, out[0] if isinstance(out, tuple) else out))
    e = float((res[1][1] - res[0][1]).norm() / res[0][1].norm())
    print("%-28s wgrad: im2col %7.1f us %5.1f TF/s | rows %7.1f us %5.1f TF/s  rel diff %.1e" % (
        (Cin, Lin, Cout, K, S), res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, e), flush=True)
    res = []
    for mode in ("0", None):
        if mode is None: os.environ.pop("MSYNTH_ROWS2", None)
        else: os.environ["MSYNTH_ROWS2"] = mode
        us, out = timeit(lambda: P.convt1d_bwd_data(gy, ya, w, d))
        res.append((us, out))
    e = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print("%-28s bwd_data: gen1 %7.1f us %5.1f TF/s | gen2 %7.1f us %5.1f TF/s  rel maxdiff %.1e" % (
        "", res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, e), flush=True)
    res = []
    bias = torch.randn(Cout, device="cuda")
    for mode in ("0", None):
        if mode is None: os.environ.pop("MSYNTH_ROWS2", None)
        else: os.environ["MSYNTH_ROWS2"] = mode
        us, out = timeit(lambda: P.convt1d_fwd(x, w, bias, d, lo))
        res.append((us, out[0] if isinstance(out, tuple) else out))
    e = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print("%-28s fwd:      gen1 %7.1f us %5.1f TF/s | gen2 %7.1f us %5.1f TF/s  rel maxdiff %.1e" % (
        "", res[0][0], fl / res[0][0] / 1e6, res[1][0], fl / res[1][0] / 1e6, e), flush=True)
