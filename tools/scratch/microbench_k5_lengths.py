import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L
import ctypes
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
C, K = 1024, 5
for B in (64, 32):
    for Lg in (8, 9, 12, 16, 17, 18, 20, 24, 28, 32, 36, 64):
        x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, pad=2, dil=1, act=1)
        gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
        fl = 2.0 * B * C * Lg * C * K
        tf = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo, want_y_act=True))
        tb = timeit(lambda: P.conv1d_bwd_data(gy, ya, w, d))
        nm = L.load().ms_conv1d_kernel_name(ctypes.byref(d), 0).decode()
        print("B=%d L=%2d fwd %6.1f us (%5.1f TF) bwd %6.1f us (%5.1f TF)  %s" % (B, Lg, tf, fl / tf / 1e6, tb, fl / tb / 1e6, nm), flush=True)
