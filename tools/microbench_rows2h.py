import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 32
for (C, Lg, dil) in ((128, 2048, 1), (128, 2048, 9), (64, 4096, 1), (64, 4096, 3)):
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    res = torch.randn(B, C, Lg, device="cuda"); gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil, dil=dil, act=1)
    line = "C=%d L=%d dil=%d " % (C, Lg, dil)
    for mode in ("0", "1"):
        os.environ["MSYNTH_ROWS2H"] = mode
        t1 = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo))
        t2 = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo, residual=res, want_y_act=True))
        t3 = timeit(lambda: P.conv1d_bwd_data(gy, ya, w, d))
        t4 = timeit(lambda: P.conv1d_bwd_data(gy, ya, w, d, gx_add=res))
        line += "| 2h=%s fwd %5.1f fwd+res+yact %5.1f bwd %5.1f bwd+add %5.1f " % (mode, t1, t2, t3, t4)
    print(line, flush=True)
