"""r04: batched atom weight gradients with operand bounds (block-scaled fp16 x 2, three products) vs without (bf16 x 3, six):
time per ResidualStack and error of both against float64, gradient-sized operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
import torch.nn.functional as F
from featuresynth._ops import prims as P, lib as L

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

torch.manual_seed(0)
B = 32
for (C, Lg) in ((256, 256), (128, 2048), (64, 4096)):
    jobs, jobs_b, refs = [], [], []
    for dil in (1, 9, 1, 3, 1, 1):
        x = torch.randn(B, C, Lg, device="cuda") * 0.3; gy = torch.randn(B, C, Lg, device="cuda") * 1e-6; ya = torch.randn(B, C, Lg, device="cuda")
        d, _ = P.conv_desc(x.shape, (C, C, 3), pad=dil, dil=dil, act=1)
        gw = torch.zeros(C, C, 3, device="cuda"); gb = torch.zeros(C, device="cuda")
        jobs.append((x, gy, ya, d, (C, C, 3), gw, gb, False))
        xm = torch.zeros(L.ATOM_AMAX_N, device="cuda"); gm = torch.zeros(L.ATOM_AMAX_N, device="cuda")
        xm[3] = x.abs().max(); gm[700] = gy.abs().max()
        gw2 = torch.zeros(C, C, 3, device="cuda"); gb2 = torch.zeros(C, device="cuda")
        jobs_b.append((x, gy, ya, d, (C, C, 3), gw2, gb2, False, xm, gm))
        if dil == 9:
            gp = torch.where(ya > 0, gy, gy * 0.2).double()
            xp = F.pad(x.double(), (dil, dil))
            ref = torch.stack([torch.einsum("bot,bit->oi", gp, xp[:, :, k * dil:k * dil + Lg]) for k in range(3)], dim=2)
            refs.append((len(jobs) - 1, ref))
    t3 = timeit(lambda: P.conv1d_bwd_weight_multi(jobs))
    t2 = timeit(lambda: P.conv1d_bwd_weight_multi(jobs_b))
    fl = 6 * 2.0 * B * C * C * 3 * Lg
    k, ref = refs[0]
    e3 = float((jobs[k][5].double() - ref).norm() / ref.norm()); e2 = float((jobs_b[k][5].double() - ref).norm() / ref.norm())
    print("C=%3d L=%4d  bf16x3 %7.1f us (%5.1f TF/s) err %.1e | fp16x2 with bounds %7.1f us (%5.1f TF/s) err %.1e" % (
        C, Lg, t3, fl / t3 / 1e6, e3, t2, fl / t2 / 1e6, e2), flush=True)
