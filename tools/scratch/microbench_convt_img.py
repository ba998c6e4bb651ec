"""ConvTranspose1d forward of the generator's four layers: image kernel (csrc/convt_img.hip) vs the row-tile kernels
(MSYNTH_CONVTIMG=0 is read once per process: the generic path is timed through ms_convt1d_fwd's own entry)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P, lib as L


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


def generic(x, w, b, d, lo):
    y = torch.empty((d.B, d.Cout, lo), dtype=torch.float32, device=x.device)
    nws = L.load().ms_convt1d_workspace_bytes(d, 0)
    ws = L.workspace(nws, x.device)
    L.call("ms_convt1d_fwd", None, d, x.data_ptr(), w.data_ptr(), L.ptr(b), y.data_ptr(), L.ptr(ws), nws, L.stream())
    return y


torch.manual_seed(0)
for B in (32, 1):
    for (Cin, Lin, Cout, K, S) in ((512, 32, 256, 16, 8), (256, 256, 128, 16, 8), (128, 2048, 64, 4, 2), (64, 4096, 32, 4, 2)):
        x = torch.randn(B, Cin, Lin, device="cuda"); w = torch.randn(Cin, Cout, K, device="cuda") * 0.02; b = torch.randn(Cout, device="cuda")
        d, lo = P.convt_desc(x.shape, w.shape, S, S // 2, act=1)
        fl = 2.0 * B * Cin * Cout * K * Lin
        t1, y1 = timeit(lambda: P.convt1d_img_fwd(x, w, b, d, lo))
        t2, y2 = timeit(lambda: generic(x, w, b, d, lo))
        print("B=%-2d %-26s image (pack + conv) %6.1f us (%5.1f TFLOP/s) | row-tile %6.1f us | x%.2f | rel diff %.1e"
              % (B, (Cin, Lin, Cout, K, S), t1, fl / t1 / 1e6, t2, t2 / t1, float((y1 - y2).norm() / y2.norm())), flush=True)
