"""r04 probe: the fused ResidualStack launch vs its three atom launches at the bench shapes (B = 32) and at B = 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import graph as G, prims as P


def timeit(fn, n=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
for B in (32, 1):
    for (C, Lg) in [(32, 8192), (64, 4096), (128, 2048)]:
        x = torch.randn(B, C, Lg, device="cuda")
        ws = [(torch.randn(C, C, 3, device="cuda") / (3 * C) ** 0.5, torch.randn(C, device="cuda") * 0.1,
               torch.randn(C, C, 3, device="cuda") / (3 * C) ** 0.5, torch.randn(C, device="cuda") * 0.1) for _ in range(3)]
        imgs = [P.atom_image(C, x.device) for _ in ws]
        P.atom_pack([(w[0], w[2], im) for w, im in zip(ws, imgs)])

        def atoms():
            h = x
            for w, im, d in zip(ws, imgs, (1, 3, 9)):
                h, _ = G.atom_forward(h, w[0], w[1], w[2], w[3], d, False, image=im)
            return h

        def stack():
            return P.stack_fwd(x, imgs, [w[1] for w in ws], [w[3] for w in ws], (1, 3, 9))
        ya, ys = atoms(), stack()
        err = float((ya.double() - ys.double()).norm() / ya.double().norm())
        print("B=%-2d C=%-3d L=%-4d three atoms %6.1f us | stack %6.1f us | rel %.1e" % (B, C, Lg, timeit(atoms), timeit(stack), err), flush=True)
