import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch, ctypes
from featuresynth._ops import prims as P, lib as L
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (64, 32):
    for Lg in (32, 17, 9):
        x = torch.randn(B, 1024, Lg, device="cuda"); w = torch.randn(1, 1024, 3, device="cuda") * 0.02; b = torch.randn(1, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, pad=1)
        gy = torch.randn(B, 1, Lg, device="cuda")
        tf = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo))
        tb = timeit(lambda: P.conv1d_bwd_data(gy, None, w, d))
        tw = timeit(lambda: P.conv1d_bwd_weight(x, gy, None, d, w.shape))
        print("B=%d L=%2d judge fwd %6.1f us  bwd_data %6.1f us  bwd_weight %6.1f us  %s" % (B, Lg, tf, tb, tw, L.load().ms_conv1d_kernel_name(ctypes.byref(d), 0).decode()), flush=True)
