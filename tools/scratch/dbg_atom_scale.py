import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
import torch.nn.functional as F
from featuresynth._ops import graph as G, prims as P
def dev(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
for wexp in (-18, -10, 0, 12):
    ws = 2.0 ** wexp
    rng = np.random.default_rng(100 + wexp)
    C, Lg = 64, 512
    x = (rng.standard_normal((2, C, Lg)) / ws).astype(np.float32)
    w0 = (rng.standard_normal((C, C, 3)) / np.sqrt(3 * C) * ws).astype(np.float32)
    w1 = (rng.standard_normal((C, C, 3)) / np.sqrt(3 * C) * ws).astype(np.float32)
    b0 = (rng.standard_normal(C) * 0.1).astype(np.float32)
    b1 = (rng.standard_normal(C) * 0.1 / ws).astype(np.float32)
    xt, w0t, w1t, b0t, b1t = (dev(a) for a in (x, w0, w1, b0, b1))
    img, imgb = P.atom_image(C, xt.device), P.atom_image(C, xt.device); P.atom_pack([(w0t, w1t, img)]); P.atom_pack([(w0t, w1t, imgb)], backward=True)
    y, rec = G.atom_forward(xt, w0t, b0t, w1t, b1t, 3, True, image=img)
    xd = torch.from_numpy(x).double()
    t = F.leaky_relu(F.conv1d(xd, torch.from_numpy(w0).double(), torch.from_numpy(b0).double(), padding=3, dilation=3), 0.2)
    u = F.leaky_relu(F.conv1d(t, torch.from_numpy(w1).double(), torch.from_numpy(b1).double(), padding=1), 0.2)
    tail = img.view(torch.float32)[-1:].cpu()
    ud = rec[4].double().cpu()
    print("wexp", wexp, "u ref absmax %.3g dev absmax %.3g rel %.3g | y-x rel %.3g | finite %s" % (
        float(u.abs().max()), float(ud.abs().max()), float((ud - u).norm() / u.norm()),
        float(((y - xt).double().cpu() - u).norm() / u.norm()), bool(torch.isfinite(y).all())))
    print("   u[0,:4,10] ref", u[0, :4, 10].numpy(), "dev", ud[0, :4, 10].numpy())
