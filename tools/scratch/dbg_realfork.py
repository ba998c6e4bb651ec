"""RealMelGan, forked discriminators under hipGraph replay: which gradients differ from the serial schedule?
One process per setting (MSYNTH_REAL_FORK=0/1); prints per-parameter max |diff| of the D / G gradient buckets after the
replayed calls against the eager (MSYNTH_GRAPH=0) run of the same setting."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import featuresynth as fs
from featuresynth import loss as LS
from featuresynth._synthetic import synthetic_features, synthetic_samples
from featuresynth.experiment import realmelgan as R
from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
import test_gpu_realmelgan as T

def run(graph):
    os.environ["MSYNTH_GRAPH"] = graph
    g, d, _, _ = T._nets()
    go = fs.FlatAdam(g.parameters(), lr=0.0, betas=(0.5, 0.9)); do = fs.FlatAdam(d.parameters(), lr=0.0, betas=(0.5, 0.9))
    dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss); gt = GeneratorTrainer(g, go, d, do, R.mel_gan_gen_loss)
    s = torch.from_numpy(synthetic_samples(2, 1024, rank=1)).cuda(); f = torch.from_numpy(synthetic_features(2, 128, 4, rank=1)).cuda()
    out = []
    for i in range(6):
        r = dt.train(s, f) if i % 2 == 0 else gt.train(s, f)
        torch.cuda.synchronize()
        opt, net = (do, d) if i % 2 == 0 else (go, g)
        out.append((r.get("d_loss", r.get("g_loss")), {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()}))
    return out
a = run("0"); b = run("1")
for i in range(6):
    worst = sorted(((float(np.abs(a[i][1][k] - b[i][1][k]).max()), float(np.abs(a[i][1][k]).max()), k) for k in a[i][1]), reverse=True)[:5]
    print("call %d loss eager %.6f graph %.6f  worst grad diffs:" % (i, a[i][0], b[i][0]), [(("%.2e" % w[0]), ("%.2e" % w[1]), w[2]) for w in worst])
