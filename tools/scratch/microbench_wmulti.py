"""Batched weight gradients (ms_conv1d_bwd_weight_multi) vs the same jobs one by one, per ResidualStack."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import prims as P

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

torch.manual_seed(0)
B = 32
for (C, Lg) in ((256, 256), (128, 2048), (64, 4096), (32, 8192)):
    jobs = []
    for dil in (1, 9, 1, 3, 1, 1):
        x = torch.randn(B, C, Lg, device="cuda"); gy = torch.randn(B, C, Lg, device="cuda"); ya = torch.randn(B, C, Lg, device="cuda")
        d, _ = P.conv_desc(x.shape, (C, C, 3), pad=dil, dil=dil, act=1)
        gw = torch.zeros(C, C, 3, device="cuda"); gb = torch.zeros(C, device="cuda")
        jobs.append((x, gy, ya, d, (C, C, 3), gw, gb, False))
    t_multi = timeit(lambda: P.conv1d_bwd_weight_multi(jobs))
    t_single = timeit(lambda: [P.conv1d_bwd_weight(*j) for j in jobs])
    fl = 6 * 2.0 * B * C * C * 3 * Lg
    print("C=%3d L=%4d  six single calls %7.1f us (%5.1f TF/s) | batched %7.1f us (%5.1f TF/s)" % (
        C, Lg, t_single, fl / t_single / 1e6, t_multi, fl / t_multi / 1e6), flush=True)
