import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, C, Lg, K, dil) in ((64, 1024, 32, 5, 1), (32, 1024, 32, 5, 1), (64, 1024, 17, 5, 1), (32, 1024, 17, 5, 1), (64, 1024, 9, 5, 1), (32, 1024, 9, 5, 1)):
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    res = torch.randn(B, C, Lg, device="cuda"); gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    fl = 2.0 * B * C * Lg * C * K
    for gen in ("0", "1"):
        os.environ["MSYNTH_ROWS2"] = gen
        msg = "%s gen%s:" % ((B, C, Lg, K, dil), "2" if gen == "1" else "1")
        for cfg in ("9",):
            os.environ["MSYNTH_ROWCFG"] = cfg
            a = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo, residual=res, want_y_act=True))
            a2 = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo))
            c = timeit(lambda: P.conv1d_bwd_data(gy, ya, w, d, gx_add=res))
            msg += "  cfg%s fwd+res %.1f fwd %.1f bwd %.1f us" % (cfg, a, a2, c)
        print(msg, flush=True)
