"""The discriminator's 1024->1024 k5 convolutions at L = 32 / 17 / 9: tile shape sweep (MSYNTH_ROWCFG)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    if isinstance(out, tuple): out = out[0]
    return e0.elapsed_time(e1) / n * 1e3, out

torch.manual_seed(0)
C, K = 1024, 5
for B in (32, 64):
    for Lg in (32, 17, 9):
        x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, pad=2, dil=1, act=1)
        gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
        fl = 2.0 * B * C * Lg * C * K
        line = "B=%d L=%2d " % (B, Lg)
        ref = None
        for cfg in (None, "0", "1", "2"):
            if cfg is None: os.environ.pop("MSYNTH_ROWCFG", None)
            else: os.environ["MSYNTH_ROWCFG"] = cfg
            tf, yf = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo, want_y_act=True))
            tb, yb = timeit(lambda: P.conv1d_bwd_data(gy, ya, w, d))
            if ref is None: ref = (yf, yb)
            ef = float((yf - ref[0]).abs().max() / ref[0].abs().max()); eb = float((yb - ref[1]).abs().max() / ref[1].abs().max())
            line += "| cfg %s fwd %6.1f us (%5.1f TF) bwd %6.1f us (%5.1f TF) d %.0e %.0e " % (cfg or "-", tf, fl / tf / 1e6, tb, fl / tb / 1e6, ef, eb)
        print(line, flush=True)
os.environ.pop("MSYNTH_ROWCFG", None)
