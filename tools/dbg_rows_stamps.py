import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
from featuresynth._ops import prims as P, lib as L
lib = L.load()
for (B, C, Lg, K, dil) in ((32, 128, 2048, 3, 1), (32, 256, 256, 3, 1), (64, 1024, 32, 5, 1)):
    x = torch.randn(B, C, Lg, device="cuda"); w = torch.randn(C, C, K, device="cuda") * 0.02; b = torch.randn(C, device="cuda")
    d, lo = P.conv_desc(x.shape, w.shape, pad=dil * (K - 1) // 2, dil=dil, act=1)
    for _ in range(3): P.conv1d_fwd(x, w, b, d, lo)
    torch.cuda.synchronize()
    st = np.zeros(64 * 6, np.int64)
    rc = lib.ms_dbg_stamps(st.ctypes.data_as(ctypes.c_void_p))
    st = st.reshape(64, 6)
    t0 = st[:, 0].min()
    print((B, C, Lg, K, dil), "rc", rc, "per WG (us): start, prologue, loop, epilogue | GHz | chunks")
    for r in st[:6]:
        print("   start %.1f  prologue %.1f  loop %.1f  epilogue %.1f | %.2f | %d" % ((r[0] - t0) / 100, (r[1] - r[0]) / 100, (r[2] - r[1]) / 100, (r[3] - r[2]) / 100, r[4] / max(1, (r[2] - r[1]) * 10), r[5]))
    print("   all WGs sampled: last end %.1f us" % ((st[:, 3].max() - t0) / 100))
