"""The k5 layer over the three scales (parts launches): forward, backward data (B and B/2 rows), weight gradient.
usage: [MSYNTH_LIB=other.so] python tools/scratch/microbench_k5_parts.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
C, B = 1024, 64
w = torch.randn(C, C, 5, device="cuda") * 0.02; b = torch.randn(C, device="cuda") * 0.1
xs = [torch.randn(B, C, l, device="cuda") for l in (32, 17, 9)]
d, lo = P.conv_desc(xs[0].shape, w.shape, pad=2, act=1)
img_f, img_b = P.conv_img_pack(d, w), P.conv_img_pack(d, w, backward=True)
ys = P.conv1d_parts_fwd(xs, w, b, d, image=img_f)
fl = sum(2.0 * B * C * C * 5 * l for l in (32, 17, 9))
for rows in (64, 32):
    gys = [torch.randn(rows, C, l, device="cuda") for l in (32, 17, 9)]
    adds = [torch.randn(rows, C, l, device="cuda") for l in (32, 17, 9)]
    t1 = timeit(lambda: P.conv1d_parts_fwd(xs, w, b, d, image=img_f)) if rows == 64 else 0.0
    t2 = timeit(lambda: P.conv1d_parts_bwd_data(gys, ys, w, d, [x.shape for x in xs], gx_adds=adds, image_bwd=img_b))
    gw, gb = P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape)
    t3 = timeit(lambda: P.conv1d_parts_bwd_weight(xs, gys, ys, d, w.shape, gw, gb, False))
    f = fl * rows / 64
    print("%s rows=%d: fwd %6.1f us (%5.1f TFLOP/s) | bwd data %6.1f us (%5.1f) | wgrad %6.1f us (%5.1f)" % (
        os.environ.get("MSYNTH_LIB", "default"), rows, t1, fl / t1 / 1e6 if t1 else 0, t2, f / t2 / 1e6, t3, f / t3 / 1e6), flush=True)
