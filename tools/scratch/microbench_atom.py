"""Fused ResidualAtom forward (one launch, csrc/atom_fused.hip) vs the two row-tile launches, at the generator's
shapes (B = 32 and B = 1), inference and training (saved activations) mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import graph as G, prims as P


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
Bs = [int(v) for v in sys.argv[1:]] or [32, 1]
for B in Bs:
    for (C, Lg) in ((32, 8192), (64, 4096), (128, 2048), (256, 256)):
        for dil in (1, 9):
            x = torch.randn(B, C, Lg, device="cuda")
            w0 = torch.randn(C, C, 3, device="cuda") * 0.05; w1 = torch.randn(C, C, 3, device="cuda") * 0.05
            b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
            img = P.atom_image(C, x.device)
            t_pack = timeit(lambda: P.atom_pack([(w0, w1, img)]))
            fl = 2 * 2.0 * B * C * Lg * C * 3
            nb = 4.0 * x.numel()
            for save in (False, True):
                tf = timeit(lambda: G.atom_forward(x, w0, b0, w1, b1, dil, save, image=img))
                tu = timeit(lambda: G.atom_forward(x, w0, b0, w1, b1, dil, save, image=None))
                by = nb * (2 + 2 * save)
                print("B=%-2d C=%-3d L=%-4d dil=%d %-9s fused %6.1f us (%5.1f TFLOP/s, %4.2f TB/s on x+y%s) | two launches %6.1f us | x%.2f | pack %.1f us"
                      % (B, C, Lg, dil, "training" if save else "inference", tf, fl / tf / 1e6, by / tf / 1e6,
                         "+t+u" if save else "", tu, tu / tf, t_pack), flush=True)
print("-- backward data")
for (C, Lg) in ((32, 8192), (64, 4096), (128, 2048), (256, 256)):
    for dil in (1, 3, 9):
        B = 32
        if not P.atom_bwd_supported(B, C, Lg, dil):
            continue
        x = torch.randn(B, C, Lg, device="cuda")
        w0 = torch.randn(C, C, 3, device="cuda") * 0.05; w1 = torch.randn(C, C, 3, device="cuda") * 0.05
        b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
        y, rec = G.atom_forward(x, w0, b0, w1, b1, dil, True, image=None)
        d0, d1, _, t, u = rec
        g = torch.randn_like(x)
        img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)], backward=True)
        tf = timeit(lambda: P.atom_bwd_data(g, u, t, img, dil))
        def two():
            gt = P.conv1d_bwd_data(g, u, w1, d1)
            return P.conv1d_bwd_data(gt, t, w0, d0, gx_add=g)
        tu = timeit(two)
        print("B=%d C=%-3d L=%-4d dil=%d backward fused %6.1f us | two launches %6.1f us | x%.2f" % (B, C, Lg, dil, tf, tu, tu / tf), flush=True)
