"""Stride-2 transposed-conv weight gradient on short rows: csrc/wgrad_convt2s.hip vs the generic kernels."""
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
    x = torch.randn(B, Cin, W, device="cuda"); gy = torch.randn(B, Cout, 2 * W, device="cuda"); y = torch.randn_like(gy)
    d, _ = P.convt_desc(x.shape, (Cin, Cout, 4), 2, 1, act=1)
    flops = 2.0 * B * W * Cin * Cout * 4
    os.environ.pop("MSYNTH_WGRADT2S", None)
    t_new = timeit(lambda: P.convt1d_bwd_weight(x, gy, y, d, (Cin, Cout, 4)))
    os.environ["MSYNTH_WGRADT2S"] = "0"
    t_old = timeit(lambda: P.convt1d_bwd_weight(x, gy, y, d, (Cin, Cout, 4)))
    os.environ.pop("MSYNTH_WGRADT2S", None)
    print("rows %5d %4d <- %3d W %2d: generic %7.1f us (%5.1f TF/s)   short-row GEMM %7.1f us (%5.1f TF/s)   (bias gradient included)"
          % (B, Cin, Cout, W, t_old, flops / t_old / 1e6, t_new, flops / t_new / 1e6))
