import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MSYNTH_GRAPH"] = "0"
import numpy as np, torch
import featuresynth as fs
from featuresynth import loss as LS
from featuresynth._synthetic import synthetic_features, synthetic_samples
from featuresynth.train import GeneratorTrainer
from test_gpu_networks import make_nets, _oracle_step, dev, host
from conftest import rel_l2
B, T = 2, 8
samples, feats = synthetic_samples(B, T * 256), synthetic_features(B, 80, T)
out = {}
for mode in ("0", None):
    if mode is None: os.environ.pop("MSYNTH_SPLIT_WGS", None)
    else: os.environ["MSYNTH_SPLIT_WGS"] = mode
    g, d, gsd, dsd = make_nets(dict(seed=7, bias_scale=0.02), dict(seed=8, bias_scale=0.02))
    go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9)); do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
    tr = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss)
    res = tr.train(dev(samples), dev(feats))
    out[mode] = {k: host(p.grad).copy() for k, p in g.named_parameters()}
o_loss, o_grads, o_params, o_fake = _oracle_step("g", gsd, dsd, samples, feats)
for k in o_grads:
    a, b = out["0"][k], out[None][k]
    print("%-28s nosplit-vs-oracle %.2e  split-vs-oracle %.2e  split-vs-nosplit %.2e  maxabs diff %.2e (|g| %.2e)" % (
        k, rel_l2(a, o_grads[k]), rel_l2(b, o_grads[k]), rel_l2(b, a), np.abs(b - a).max(), np.abs(a).max()))
