"""Timing probes of the fused atom kernel (training mode): MSYNTH_ATOM_DBG = 0 product | 1 weights not streamed |
2 no B-fragment LDS reads in the K loop | 3 no global stores | 4 no x window staging | 5 no MFMAs.
  for d in 0 1 2 3 4 5; do MSYNTH_ATOM_DBG=$d python3 tools/probe_atom.py; done"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch
from featuresynth._ops import graph as G, prims as P
out = []
for (C, Lg) in ((32, 8192), (64, 4096), (128, 2048)):
    B, dil = 32, 3
    x = torch.randn(B, C, Lg, device="cuda")
    w0 = torch.randn(C, C, 3, device="cuda") * 0.05; w1 = torch.randn(C, C, 3, device="cuda") * 0.05
    b0 = torch.randn(C, device="cuda") * 0.1; b1 = torch.randn(C, device="cuda") * 0.1
    img = P.atom_image(C, x.device); P.atom_pack([(w0, w1, img)])
    fn = lambda: G.atom_forward(x, w0, b0, w1, b1, dil, True, image=img)
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
    for _ in range(40): fn()
    e1.record(); torch.cuda.synchronize()
    out.append("C=%d %.1f us" % (C, e0.elapsed_time(e1) / 40 * 1e3))
print("MSYNTH_ATOM_DBG=%s  " % os.environ.get("MSYNTH_ATOM_DBG", "0") + " | ".join(out), flush=True)
