import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L
torch.manual_seed(0)
def run(fn):
    outs = []
    for mode in ("0", None):
        if mode is None: os.environ.pop("MSYNTH_SPLIT_WGS", None)
        else: os.environ["MSYNTH_SPLIT_WGS"] = mode
        o = fn(); torch.cuda.synchronize()
        outs.append([t.clone() for t in (o if isinstance(o, tuple) else (o,)) if t is not None])
    return max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(*outs))
B = 2
for C, Lg in ((256, 64), (128, 512), (64, 1024), (32, 2048), (512, 8), (1024, 8), (1024, 3)):
    for dil in (1, 3, 9):
        x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
        res = torch.randn(B, C, Lg, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, pad=dil, dil=dil, act=1)
        e1 = run(lambda: P.conv1d_fwd(x, w, b, d, lo))
        e2 = run(lambda: P.conv1d_fwd(x, w, b, d, lo, residual=res, want_y_act=True))
        gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
        e3 = run(lambda: P.conv1d_bwd_data(gy, ya, w, d))
        e4 = run(lambda: P.conv1d_bwd_data(gy, ya, w, d, gx_add=res))
        print("C%d L%d d%d: fwd %.1e fwd+res %.1e bwd %.1e bwd+add %.1e" % (C, Lg, dil, e1, e2, e3, e4), flush=True)
