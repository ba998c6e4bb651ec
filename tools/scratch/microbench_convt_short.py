"""Stride-2 transposed-conv forward on short rows: csrc/convt_fwd_short.hip vs the generic row kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for B, Cin, W, Cout in [(128, 2048, 4, 512), (256, 1024, 8, 256), (512, 512, 16, 128), (1024, 256, 32, 128)]:
    x = torch.randn(B, Cin, W, device="cuda"); w = torch.randn(Cin, Cout, 4, device="cuda") * 0.05; b = torch.randn(Cout, device="cuda")
    d, lo = P.convt_desc(x.shape, w.shape, 2, 1, act=1)
    flops = 2.0 * B * W * Cin * Cout * 4
    os.environ.pop("MSYNTH_CONVTSHORT", None)
    t_new = timeit(lambda: P.convt1d_fwd(x, w, b, d, lo))
    os.environ["MSYNTH_CONVTSHORT"] = "0"
    t_old = timeit(lambda: P.convt1d_fwd(x, w, b, d, lo))
    os.environ.pop("MSYNTH_CONVTSHORT", None)
    print("rows %5d %4d -> %3d W %2d: generic %7.1f us (%5.1f TF/s)   short-row image kernel %7.1f us (%5.1f TF/s, pack included)"
          % (B, Cin, Cout, W, t_old, flops / t_old / 1e6, t_new, flops / t_new / 1e6))
