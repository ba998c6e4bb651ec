"""The discriminator's 1024 -> 1024 k5 conv: image kernel (csrc/conv5_img.hip) vs the generic row kernels, forward and
backward data, at the three scales and both batch sizes of the train step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
C = 1024
w = torch.randn(C, C, 5, device="cuda") * 0.02; b = torch.randn(C, device="cuda") * 0.1
for B in (64, 32):
    for Lg in (32, 17, 9):
        x = torch.randn(B, C, Lg, device="cuda"); gy = torch.randn(B, C, Lg, device="cuda"); add = torch.randn(B, C, Lg, device="cuda")
        d, lo = P.conv_desc(x.shape, w.shape, pad=2, act=1)
        img_f, img_b = P.conv_img_pack(d, w), P.conv_img_pack(d, w, backward=True)
        y = P.conv1d_img_fwd(x, img_f, b, d, lo)
        fl = 2.0 * B * C * C * 5 * Lg
        tp = timeit(lambda: P.conv_img_pack(d, w))
        t1 = timeit(lambda: P.conv1d_img_fwd(x, img_f, b, d, lo)); t2 = timeit(lambda: P.conv1d_fwd(x, w, b, d, lo))
        t3 = timeit(lambda: P.conv1d_img_bwd_data(gy, y, img_b, d, gx_add=add)); t4 = timeit(lambda: P.conv1d_bwd_data(gy, y, w, d, gx_add=add))
        print("B=%d L=%-2d fwd image %6.1f us (%5.1f TFLOP/s) | generic %6.1f us | x%.2f || bwd image %6.1f us (%5.1f) | generic %6.1f us | x%.2f || pack %.1f us"
              % (B, Lg, t1, fl / t1 / 1e6, t2, t2 / t1, t3, fl / t3 / 1e6, t4, t4 / t3, tp), flush=True)
