"""Per-parameter gradient errors of the first D step / G step against tests/golden/train.npz (scratch diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_networks import make_nets, _trainers, dev, host
from featuresynth._synthetic import strided_sample, synthetic_features, synthetic_samples
z = np.load(os.path.join(ROOT, "tests/golden/train.npz"))
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-300))
for tag in sys.argv[1:] or ["cfg3"]:
    B, T, n = [int(v) for v in z[tag + "/cfg"]]
    gkw, dkw = ((dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02)) if tag == "small" else (dict(seed=7), dict(seed=7)))
    g, d, _, _ = make_nets(gkw, dkw)
    dt, gt, _, _ = _trainers(g, d, "flat")
    r = dt.train(dev(synthetic_samples(B, T * 256, rank=0)), dev(synthetic_features(B, 80, T, rank=0)))
    print(tag, "d_loss", r["d_loss"], z[tag + "/losses"][0])
    for k, p in d.named_parameters():
        smp, ref = strided_sample(host(p.grad)), z["%s/dgrad_smp/%s" % (tag, k)]
        e = rel(smp, ref)
        line = "%-22s rel %.3e" % (k, e)
        if k.endswith("bias") and e > 1e-4:
            dlt = np.abs(smp.astype(np.float64) - ref).ravel()
            o = np.argsort(dlt)[::-1][:4]
            line += "  top |d| " + " ".join("%d:%.2e(ref %.2e)" % (i, dlt[i], ref.ravel()[i]) for i in o) + "  median |d| %.2e" % np.median(dlt)
        print(line)
