import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from featuresynth._synthetic import synthetic_features
from featuresynth._ops import graph as G
from test_gpu_networks import make_nets, dev
B, T = 2, 8
feats = dev(synthetic_features(B, 80, T))
g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
params = [p.detach() for p in g.parameters()]
tapes = {}
for mode in ("0", None):
    if mode is None: os.environ.pop("MSYNTH_SPLIT_WGS", None)
    else: os.environ["MSYNTH_SPLIT_WGS"] = mode
    y, tape = G.gen_forward(feats, params, True)
    torch.cuda.synchronize()
    tapes[mode] = tape
for ra, rb in zip(tapes["0"], tapes[None]):
    if ra[0] == "atom":
        names, ta, tb = ("t", "u"), ra[1][3:5], rb[1][3:5]
    else:
        names, ta, tb = ("out",), ra[3:4], rb[3:4]
    for n, a, b in zip(names, ta, tb):
        flip = (a > 0) != (b > 0)
        nf = int(flip.sum())
        msg = ""
        if nf:
            msg = " values at flips: %s / %s" % (a[flip][:4].tolist(), b[flip][:4].tolist())
        print(ra[0], n, tuple(a.shape), "flips", nf, "maxdiff %.2e" % float((a - b).abs().max()), msg)
